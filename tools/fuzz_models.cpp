// Host-side robustness check of the model readers and the flatteners (CPU only, built with
// -fsanitize=address,undefined by tests/test_model_io.py): mutated legacy-binary, JSON and UBJSON images
// must either load and flatten cleanly or be refused with an OhxError - never crash, never read out of
// bounds.  usage: fuzz_models <model file> <iterations> <seed>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "flatten.hpp"
#include "forest.hpp"

using namespace ohx;

static int try_image(const std::vector<uint8_t>& img) {
  try {
    Forest f = load_model_buffer(img.data(), img.size());
    f.validate();
    SuperForest sf;
    (void)emit_super(f, &sf);
    LayoutParams lp;
    Placement p = place_forest(f, lp);
    if (packed_format_fits(f, p)) (void)emit_packed(f, p, nullptr);
    (void)emit_wide(f, p);
    (void)write_legacy_binary(f);
    return 1;
  } catch (const OhxError&) {
    return 0;
  } catch (const std::bad_alloc&) {
    return 0;
  } catch (const std::length_error&) {
    return 0;
  }
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<uint8_t> base((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  const int iters = atoi(argv[2]);
  std::mt19937 rng((unsigned)atoi(argv[3]));
  // the three encodings of the same booster
  Forest f0 = load_model_buffer(base.data(), base.size());
  std::string js = write_json_model(f0);
  std::vector<std::vector<uint8_t>> seeds = {write_legacy_binary(f0), std::vector<uint8_t>(js.begin(), js.end()),
                                             write_ubjson_model(f0)};
  long accepted = 0, refused = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> img = seeds[it % seeds.size()];
    const int kind = (int)(rng() % 5);
    if (kind == 0 && !img.empty()) {
      img.resize(rng() % img.size());                                   // truncate
    } else if (kind == 1) {
      for (int k = 0, n = 1 + (int)(rng() % 8); k < n && !img.empty(); ++k) img[rng() % img.size()] ^= (uint8_t)(1u << (rng() % 8));
    } else if (kind == 2) {
      for (int k = 0, n = 1 + (int)(rng() % 4); k < n && !img.empty(); ++k) img[rng() % img.size()] = (uint8_t)rng();
    } else if (kind == 3 && img.size() > 16) {
      const size_t a = rng() % (img.size() - 8), len = 1 + rng() % 8;  // overwrite a field with an extreme value
      for (size_t q = 0; q < len && a + q < img.size(); ++q) img[a + q] = (rng() & 1) ? 0xFF : 0x00;
    } else if (img.size() > 32) {
      const size_t a = rng() % (img.size() - 16), b = rng() % (img.size() - 16);   // splice
      memcpy(&img[a], &seeds[it % seeds.size()][b], 16);
    }
    (try_image(img) ? accepted : refused)++;
  }
  printf("fuzz_models: %d images, %ld accepted, %ld refused, 0 crashes\n", iters, accepted, refused);
  return 0;
}
