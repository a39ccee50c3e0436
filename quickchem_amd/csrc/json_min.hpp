// Minimal JSON reader for XGBoost model files.  Arrays of numbers/booleans are
// kept as flat numeric vectors (tree arrays hold one entry per node), everything
// else as a small DOM.  Numbers keep both their double and their directly-parsed
// float value so a float field is rounded once, from the decimal text.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "forest.hpp"

namespace ohx {
namespace json {

struct Num {
  double d;
  float f;
};

struct Value {
  enum Type { Null, Bool, Number, String, Array, NumArray, Object } type = Null;
  bool b = false;
  Num num{0.0, 0.0f};
  std::string str;
  std::vector<Value> arr;
  std::vector<Num> nums;
  std::map<std::string, Value> obj;

  const Value& at(const std::string& key) const {
    if (type != Object) throw OhxError("JSON: expected an object while looking up '" + key + "'");
    auto it = obj.find(key);
    if (it == obj.end()) throw OhxError("JSON: missing key '" + key + "'");
    return it->second;
  }
  const Value* find(const std::string& key) const {
    if (type != Object) return nullptr;
    auto it = obj.find(key);
    return it == obj.end() ? nullptr : &it->second;
  }
  size_t array_size() const {
    if (type == NumArray) return nums.size();
    if (type == Array) return arr.size();
    throw OhxError("JSON: expected an array");
  }
};

class Parser {
 public:
  Parser(const char* text, size_t len) : p_(text), end_(text + len) {}

  Value parse() {
    Value v = value();
    ws();
    // xgboost accepts a trailing NUL (its loader checks str[size-2] == '}')
    while (p_ < end_ && (*p_ == '\0')) ++p_;
    ws();
    if (p_ != end_) fail("trailing characters after the document");
    return v;
  }

 private:
  const char* p_;
  const char* end_;

  [[noreturn]] void fail(const std::string& what) const { throw OhxError("JSON: " + what); }

  void ws() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
  }

  bool lit(const char* s) {
    size_t n = strlen(s);
    if ((size_t)(end_ - p_) >= n && memcmp(p_, s, n) == 0) {
      p_ += n;
      return true;
    }
    return false;
  }

  Num number() {
    // xgboost's writer emits the non-standard tokens NaN / Infinity / -Infinity
    if (lit("NaN")) return {std::nan(""), std::nanf("")};
    if (lit("Infinity")) return {INFINITY, INFINITY};
    if (lit("-Infinity")) return {-INFINITY, -INFINITY};
    // copy the token so strtod/strtof cannot run past the buffer
    const char* s = p_;
    while (p_ < end_ && (strchr("+-0123456789.eE", *p_) != nullptr)) ++p_;
    if (p_ == s) fail("bad number");
    char tok[64];
    size_t n = (size_t)(p_ - s);
    if (n >= sizeof(tok)) fail("number token too long");
    memcpy(tok, s, n);
    tok[n] = 0;
    char* e1 = nullptr;
    Num out;
    out.d = strtod(tok, &e1);
    if (e1 != tok + n) fail(std::string("bad number '") + tok + "'");
    out.f = strtof(tok, nullptr);
    return out;
  }

  std::string string() {
    if (p_ >= end_ || *p_ != '"') fail("expected a string");
    ++p_;
    std::string out;
    while (p_ < end_ && *p_ != '"') {
      char c = *p_++;
      if (c == '\\') {
        if (p_ >= end_) fail("bad escape");
        char e = *p_++;
        switch (e) {
          case 'n': out.push_back('\n'); break;
          case 't': out.push_back('\t'); break;
          case 'r': out.push_back('\r'); break;
          case 'b': out.push_back('\b'); break;
          case 'f': out.push_back('\f'); break;
          case 'u': {
            if (end_ - p_ < 4) fail("bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p_, 4).c_str(), nullptr, 16);
            p_ += 4;
            if (cp < 0x80) out.push_back((char)cp);
            else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
            else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
            break;
          }
          default: out.push_back(e);
        }
      } else {
        out.push_back(c);
      }
    }
    if (p_ >= end_) fail("unterminated string");
    ++p_;
    return out;
  }

  // a model document nests six deep; a file that nests without end is refused, not recursed into
  struct Depth {
    int& d;
    explicit Depth(int& dd) : d(dd) { ++d; }
    ~Depth() { --d; }
  };
  int depth_ = 0;

  Value value() {
    Depth guard(depth_);
    if (depth_ > 64) fail("nested too deeply");
    ws();
    if (p_ >= end_) fail("unexpected end of input");
    Value v;
    char c = *p_;
    if (c == '{') {
      ++p_;
      v.type = Value::Object;
      ws();
      if (p_ < end_ && *p_ == '}') { ++p_; return v; }
      for (;;) {
        ws();
        std::string k = string();
        ws();
        if (p_ >= end_ || *p_ != ':') fail("expected ':'");
        ++p_;
        v.obj.emplace(std::move(k), value());
        ws();
        if (p_ < end_ && *p_ == ',') { ++p_; continue; }
        if (p_ < end_ && *p_ == '}') { ++p_; break; }
        fail("expected ',' or '}'");
      }
      return v;
    }
    if (c == '[') {
      ++p_;
      ws();
      if (p_ < end_ && *p_ == ']') { ++p_; v.type = Value::NumArray; return v; }
      // numeric fast path: numbers and booleans only
      const char* save = p_;
      bool numeric = true;
      v.type = Value::NumArray;
      for (;;) {
        ws();
        if (p_ >= end_) fail("unterminated array");
        char d = *p_;
        if (d == 't') { if (!lit("true")) fail("bad literal"); v.nums.push_back({1.0, 1.0f}); }
        else if (d == 'f') { if (!lit("false")) fail("bad literal"); v.nums.push_back({0.0, 0.0f}); }
        else if (d == '-' || d == 'N' || d == 'I' || (d >= '0' && d <= '9')) v.nums.push_back(number());
        else { numeric = false; break; }
        ws();
        if (p_ < end_ && *p_ == ',') { ++p_; continue; }
        if (p_ < end_ && *p_ == ']') { ++p_; break; }
        fail("expected ',' or ']'");
      }
      if (numeric) return v;
      // general array: restart
      p_ = save;
      v.nums.clear();
      v.type = Value::Array;
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (p_ < end_ && *p_ == ',') { ++p_; continue; }
        if (p_ < end_ && *p_ == ']') { ++p_; break; }
        fail("expected ',' or ']'");
      }
      return v;
    }
    if (c == '"') { v.type = Value::String; v.str = string(); return v; }
    if (c == 't') { if (!lit("true")) fail("bad literal"); v.type = Value::Bool; v.b = true; return v; }
    if (c == 'f') { if (!lit("false")) fail("bad literal"); v.type = Value::Bool; v.b = false; return v; }
    if (c == 'n') { if (!lit("null")) fail("bad literal"); v.type = Value::Null; return v; }
    if (c == '-' || c == 'N' || c == 'I' || (c >= '0' && c <= '9')) { v.type = Value::Number; v.num = number(); return v; }
    fail(std::string("unexpected character '") + c + "'");
  }
};

}  // namespace json
}  // namespace ohx
