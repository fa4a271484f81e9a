// Probe (GPU box): does the range check of a raw buffer load (stride 0) include the scalar offset?
// Prints what a buffer_load_dword and a buffer_load ... lds deliver for voffset in range / soffset pushing the
// address past num_records.   hipcc --offload-arch=gfx950 tools/probes/buf_range.hip -o /tmp/buf_range && /tmp/buf_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const unsigned* src, unsigned* out, int records, int soff) {
    __shared__ unsigned lds[64 * 4];
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
    const int lane = threadIdx.x;
    lds[lane * 4] = 0xdeadbeef;
    __syncthreads();
    out[lane] = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, soff, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, lane * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[64 + lane] = lds[lane * 4];
}
int main() {
    const int n = 4096;
    unsigned *src, *out;
    hipMalloc(&src, n * 4);
    hipMalloc(&out, 128 * 4);
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 0x1000 + i;
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int soff : {0, 512, 1024}) {
        // records = 1024 bytes (256 dwords): lanes' voffset 0..252 (dword) / 0..1008 (lds, 16 B) are in range alone
        k<<<1, 64>>>(src, out, 1024, soff);
        std::vector<unsigned> o(128);
        hipMemcpy(o.data(), out, 128 * 4, hipMemcpyDeviceToHost);
        printf("soffset %4d: dword lane0 %x lane63 %x | lds lane0 %x lane31 %x lane32 %x lane63 %x\n", soff, o[0], o[63], o[64],
               o[64 + 31], o[64 + 32], o[64 + 63]);
    }
    printf("expect in range: dword lane L = 0x%x + L + soff/4 ; lds lane L = 0x%x + 4L + soff/4\n", 0x1000, 0x1000);
    return 0;
}
