// GPU-box probe (round 6): what `buffer_load_dwordx4 ... lds` does (a) for lanes whose offset is outside the buffer
// descriptor (zeros written, or nothing?) and (b) with an LDS destination above 64 KiB.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ __launch_bounds__(256) void k(const unsigned* A, int nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned lds[160 * 256 - 256];     // 159 KiB
  const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int i = t; i < 160 * 256 - 256; i += 256) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, nbytes, 0x00020000);
  // piece 0: in range for lanes with offset < nbytes, OOB for the rest; LDS base 0 + wave KiB
  unsigned vo = (unsigned)t * 16u;
  if ((t & 7) == 3) vo = 0xC0000000u;                 // the sentinel the contraction kernels use
  if ((t & 7) == 5) vo = (unsigned)(-64 + t);          // "negative" offsets (conv taps above row 0)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)((char*)lds + wave * 1024), 16, vo, 0, 0, 0);
  // piece 1: LDS destination at 128 KiB + wave KiB
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)((char*)lds + 131072 + wave * 1024), 16, (unsigned)t * 16u, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int i = t; i < 1024; i += 256) out[i] = lds[i];
  for (int i = t; i < 1024; i += 256) out[1024 + i] = lds[32768 + i];
}
int main() {
  const int n = 4096;   // bytes in the descriptor; 256 lanes x 16 B = 4096: all in range except the patched lanes
  std::vector<unsigned> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x1000000u + i;
  unsigned *A, *out;
  hipMalloc(&A, n); hipMalloc(&out, 2048 * 4);
  hipMemcpy(A, h.data(), n, hipMemcpyHostToDevice);
  for (int nb : {4096, 2048}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, A, nb, out);
    std::vector<unsigned> o(2048);
    hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_oob = 0, stale_oob = 0, other = 0, hi_ok = 0;
    for (int t = 0; t < 256; ++t) for (int d = 0; d < 4; ++d) {
      unsigned v = o[t * 4 + d];
      bool oob = ((t & 7) == 3) || ((t & 7) == 5) || (t * 16 >= nb);
      if (!oob) { if (v == 0x1000000u + t * 4 + d) ++ok_in; else ++other; }
      else { if (v == 0) ++zero_oob; else if (v == 0xDEADBEEFu) ++stale_oob; else ++other; }
      unsigned w = o[1024 + t * 4 + d];
      bool oob2 = t * 16 >= nb;
      if (oob2 ? (w == 0) : (w == 0x1000000u + t * 4 + d)) ++hi_ok;
    }
    printf("num_records=%d: in-range ok %d, OOB->zero %d, OOB->stale(not written) %d, other %d; LDS@128KiB ok %d/1024\n",
           nb, ok_in, zero_oob, stale_oob, other, hi_ok);
  }
  return 0;
}
