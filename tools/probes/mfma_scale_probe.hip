// Which lane's scale byte applies to which k-bytes of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8)?
// For every (data lane group gd, 16-byte half h) only those A bytes are 1.0 (B all ones); then the scale of ONE lane group gs is
// doubled: the row sums double iff that group's scale covers those bytes.  Same for the B side.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const int* a, const int* b, const int* sa, const int* sb, float* c) {
  const int l = threadIdx.x;
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa[l], 0, sb[l]);
  for (int i = 0; i < 4; ++i) c[l * 4 + i] = acc[i];
}

int main() {
  int *a, *b, *sa, *sb; float* c;
  hipMalloc(&a, 64 * 32); hipMalloc(&b, 64 * 32); hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&c, 1024);
  unsigned char ha[64 * 32], hb[64 * 32]; int hsa[64], hsb[64]; float hc[256];
  for (int side = 0; side < 2; ++side) {
    printf("%s side: rows = (data lane group, 16-byte half), columns = lane group whose scale is doubled; entry = max output / 16\n", side ? "B" : "A");
    for (int gd = 0; gd < 4; ++gd)
      for (int h = 0; h < 2; ++h) {
        printf("  data group %d half %d:", gd, h);
        for (int gs = 0; gs < 4; ++gs) {
          memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
          unsigned char* sparse = side ? hb : ha; unsigned char* dense = side ? ha : hb;
          for (int l = 0; l < 64; ++l) {
            for (int i = 0; i < 32; ++i) dense[l * 32 + i] = 0x38;
            if ((l >> 4) == gd) for (int i = 0; i < 16; ++i) sparse[l * 32 + h * 16 + i] = 0x38;
          }
          for (int l = 0; l < 64; ++l) { hsa[l] = 127; hsb[l] = 127; }
          for (int l = 0; l < 64; ++l) if ((l >> 4) == gs) (side ? hsb : hsa)[l] = 128;
          hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
          hipMemcpy(sa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(sb, hsb, 256, hipMemcpyHostToDevice);
          hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, sa, sb, c);
          hipMemcpy(hc, c, 1024, hipMemcpyDeviceToHost);
          float mx = 0; for (int i = 0; i < 256; ++i) mx = hc[i] > mx ? hc[i] : mx;
          printf(" %4.1f", mx / 16);
        }
        printf("\n");
      }
  }
  // second question: does byte 0 of the lane's scale register apply (opsel 0), and is scale = 2^(s-127)?
  return 0;
}
