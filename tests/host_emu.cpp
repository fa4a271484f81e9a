// Host harness for the format / dequant helpers shared with the HIP kernels
// (mxq_amd/csrc/mxq_format.h, mxq_pack.h, mxq_dequant.h compiled for the CPU with a
// software v_perm_b32).  TEST INFRASTRUCTURE: lets `pytest -m "not gpu"` check the bit
// layout, the integer unpack and the LUT dequant against the oracle without a GPU.
//   g++ -O2 -mf16c -ffp-contract=off -shared -fPIC tests/host_emu.cpp -o tests/_build/libhost_emu.so
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

uint16_t mxq_host_f32_to_f16(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }

#include "../mxq_amd/csrc/mxq_dequant.h"
#include "../mxq_amd/csrc/mxq_format.h"
#include "../mxq_amd/csrc/mxq_pack.h"

extern "C" {

long emu_qweight_dwords(int N, int K) { return (long)(N / 16) * (K / 64) * MXQ_BLK_DW; }

void emu_pack(const uint8_t* codes2, const uint8_t* sc2, const float* zero2, const float* qs2, const float* qz2,
              const uint8_t* codes4, const uint8_t* sc4, const float* zero4, const float* qs4, const float* qz4,
              uint32_t* qweight, float* rowmeta, int N, int K) {
    const int NC = K / 64;
    memset(qweight, 0, emu_qweight_dwords(N, K) * 4);
    for (int n = 0; n < N; ++n) {
        for (int c = 0; c < NC; ++c) {
            uint32_t* tile = qweight + mxq_blk_index(n, c, K) * MXQ_BLK_DW;
            mxq_pack_row_chunk(tile, n & 15, codes2 + (long)n * NC * 48 + c * 48, sc2 + (long)n * NC * 3 + c * 3,
                               zero2 + (long)n * NC * 3 + c * 3, codes4 + (long)n * NC * 16 + c * 16);
            if ((n & 15) == 0)
                for (int g = 0; g < 3; ++g) {
                    memcpy(tile + mxq_qq(g), qs2 + (long)(n / 16) * NC * 3 + c * 3 + g, 4);
                    memcpy(tile + mxq_qq(g) + 1, qz2 + (long)(n / 16) * NC * 3 + c * 3 + g, 4);
                }
        }
        rowmeta[4 * n] = zero4[n];
        rowmeta[4 * n + 1] = (float)sc4[n];
        rowmeta[4 * n + 2] = qs4[n / 16];
        rowmeta[4 * n + 3] = qz4[n / 16];
    }
}

void emu_unpack(const uint32_t* qweight, uint8_t* codes2, uint8_t* sc2, float* zero2, uint8_t* codes4, int N, int K) {
    const int NC = K / 64;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < NC; ++c)
            mxq_unpack_row_chunk(qweight + mxq_blk_index(n, c, K) * MXQ_BLK_DW, n & 15,
                                 codes2 + (long)n * NC * 48 + c * 48, sc2 + (long)n * NC * 3 + c * 3,
                                 zero2 + (long)n * NC * 3 + c * 3, codes4 + (long)n * NC * 16 + c * 16);
}

// dequantise with exactly the per-thread code of mxq_dequant_f16_kernel / the GEMM staging
void emu_dequant_f16(const uint32_t* qweight, const float* rowmeta, uint16_t* out, int N, int K) {
    const int NC = K / 64;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < NC; ++c) {
            const uint32_t* tile = qweight + mxq_blk_index(n, c, K) * MXQ_BLK_DW;
            const int r = n & 15;
            uint32_t o[8];
            const uint32_t scw = ((const uint16_t*)tile)[mxq_sc_u16(r)];
            for (int g = 0; g < 3; ++g) {
                float qs, qz, z;
                memcpy(&qs, tile + mxq_qq(g), 4);
                memcpy(&qz, tile + mxq_qq(g) + 1, 4);
                memcpy(&z, tile + mxq_z2(g, r), 4);
                mxq_deq2x16(tile[mxq_c2(g, r)], mxq_scale(qs, qz, (scw >> (4 * g)) & 15u), z, o);
                memcpy(out + (long)n * K + c * 64 + g * 16, o, 32);
            }
            const float* m = rowmeta + 4 * n;
            const float s4 = mxq_scale(m[2], m[3], (uint32_t)m[1]);
            mxq_deq4x8(tile[mxq_c4(0, r)], s4, m[0], o);
            mxq_deq4x8(tile[mxq_c4(1, r)], s4, m[0], o + 4);
            memcpy(out + (long)n * K + c * 64 + 48, o, 32);
        }
}
}
