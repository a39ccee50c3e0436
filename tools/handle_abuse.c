/* Handle misuse against the C ABI, for an AddressSanitizer host build of libohxgb (tests/test_abi.py):
 * double frees, stale handles after the allocator has recycled the block, pointers that were never
 * handles.  Every call must come back with -1 and a message; ASan must see no use-after-free.  Needs no GPU:
 * booster handles are host objects, and a DMatrix cannot be created without a device. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/ohxgb.h"

#define EXPECT(cond)                                                    \
  do {                                                                  \
    if (!(cond)) {                                                      \
      fprintf(stderr, "handle_abuse: line %d: %s\n", __LINE__, #cond);  \
      return 1;                                                         \
    }                                                                   \
  } while (0)

int main(void) {
  BoosterHandle b = NULL, again[64];
  bst_ulong n = 0, info[8];
  const float* res = NULL;
  char stack_words[256];
  memset(stack_words, 0x4F, sizeof stack_words);

  EXPECT(XGBoosterCreate(NULL, 0, &b) == 0 && b != NULL);
  EXPECT(XGBoosterSetParam(b, "ohx_kernel", "wide") == 0);
  EXPECT(XGBoosterFree(b) == 0);
  EXPECT(XGBoosterFree(b) == -1);                                  /* double free */
  EXPECT(strstr(XGBGetLastError(), "invalid or has been freed") != NULL);
  /* let the allocator hand the block out again, then use the stale handle every way the ABI allows */
  for (int i = 0; i < 64; ++i) EXPECT(XGBoosterCreate(NULL, 0, &again[i]) == 0);
  for (int i = 0; i < 64; i += 2) EXPECT(XGBoosterFree(again[i]) == 0);
  for (int i = 0; i < 64; i += 2) {
    EXPECT(XGBoosterSetParam(again[i], "ohx_kernel", "auto") == -1);
    EXPECT(XGBoosterLoadModel(again[i], "/nonexistent") == -1);
    EXPECT(XGBoosterSaveModel(again[i], "/tmp/x") == -1);
    EXPECT(OHXBoosterGetInfo(again[i], info) == -1);
    EXPECT(XGBoosterPredict(again[i], NULL, 0, 0, 0, &n, &res) == -1);
    EXPECT(XGBoosterFree(again[i]) == -1);
  }
  for (int i = 1; i < 64; i += 2) EXPECT(XGBoosterFree(again[i]) == 0);
  /* things that never were handles */
  EXPECT(XGBoosterFree(NULL) == -1);
  EXPECT(XGBoosterFree(stack_words) == -1);
  EXPECT(XGDMatrixFree(stack_words) == -1);
  EXPECT(XGDMatrixFree(NULL) == -1);
  EXPECT(XGDMatrixNumRow(stack_words, &n) == -1);
  EXPECT(OHXDMatrixSetGrid(stack_words, 4, 4, 0) == -1);
  /* a live booster is not a DMatrix and the other way round */
  EXPECT(XGBoosterCreate(NULL, 0, &b) == 0);
  EXPECT(XGDMatrixFree(b) == -1);
  EXPECT(XGDMatrixNumCol(b, &n) == -1);
  EXPECT(XGBoosterCreate((const DMatrixHandle*)&b, 1, &again[0]) == -1);   /* len = 1: the array IS read, and b is no DMatrix */
  EXPECT(XGBoosterFree(b) == 0);
  printf("handle_abuse: ok\n");
  return 0;
}
