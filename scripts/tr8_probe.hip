#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
// LDS filled with byte value = (row * 16 + col) & 255 ... we record for each lane the 8 bytes it receives
__global__ void k(uint8_t* out, int ld) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint8_t)((i / ld) * 16 + (i % ld) % 16);  // value = row*16 + col%16 (row < 16)
  __syncthreads();
  const int lane = threadIdx.x;
  // lane supplies address of row (lane / ?) ... try: each lane points at row = lane % 16 (within 16 lanes), col base = 0
  typedef __attribute__((address_space(3))) v2i lds_v2i;
  const uint8_t* p = lds + (lane & 15) * ld + (lane >> 4) * 16;
  v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(p));
  reinterpret_cast<v2i*>(out)[lane] = r;
}
int main() {
  uint8_t* d; hipMalloc(&d, 64 * 8);
  for (int ld : {16, 64}) {
    k<<<1, 64>>>(d, ld);
    uint8_t h[512]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("ld=%d (value = row*16 + col, lane address = row lane%%16, byte col 16*(lane/16))\n", ld);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" r%02d c%02d |", h[l*8+j] >> 4, h[l*8+j] & 15); printf("\n"); }
  }
  return 0;
}
