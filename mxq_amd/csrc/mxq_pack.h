// Per-(row, chunk) pack / unpack of the lossless MXQ parameterisation into format v1.
// Host+device inline so the host harness (tests/host_emu.cpp) exercises the same code.
#pragma once
#include "mxq_format.h"

// codes2: 48 two-bit codes (3 groups x 16), sc: 3 scale codes, z: 3 zero-points,
// codes4: 16 four-bit codes.  `tile` points at the 144-dword block of (row / 16, chunk).
MXQ_HD void mxq_pack_row_chunk(uint32_t* tile, int r, const uint8_t* codes2,
                               const uint8_t* sc, const float* z, const uint8_t* codes4) {
    for (int g = 0; g < 3; ++g) {
        uint32_t w = 0;
        for (int k = 0; k < 16; ++k) w |= (uint32_t)(codes2[g * 16 + k] & 3u) << mxq_bit2(k);
        tile[mxq_c2(g, r)] = w;
        tile[mxq_z2(g, r)] = __builtin_bit_cast(uint32_t, z[g]);
    }
    for (int h = 0; h < 2; ++h) {
        uint32_t w = 0;
        for (int k = 0; k < 8; ++k) w |= (uint32_t)(codes4[h * 8 + k] & 15u) << mxq_bit4(k);
        tile[mxq_c4(h, r)] = w;
    }
    ((uint16_t*)tile)[mxq_sc_u16(r)] =
        (uint16_t)((sc[0] & 15u) | ((sc[1] & 15u) << 4) | ((sc[2] & 15u) << 8));
}

MXQ_HD void mxq_unpack_row_chunk(const uint32_t* tile, int r, uint8_t* codes2,
                                 uint8_t* sc, float* z, uint8_t* codes4) {
    for (int g = 0; g < 3; ++g) {
        const uint32_t w = tile[mxq_c2(g, r)];
        for (int k = 0; k < 16; ++k) codes2[g * 16 + k] = (uint8_t)((w >> mxq_bit2(k)) & 3u);
        z[g] = __builtin_bit_cast(float, tile[mxq_z2(g, r)]);
    }
    for (int h = 0; h < 2; ++h) {
        const uint32_t w = tile[mxq_c4(h, r)];
        for (int k = 0; k < 8; ++k) codes4[h * 8 + k] = (uint8_t)((w >> mxq_bit4(k)) & 15u);
    }
    const uint32_t s = ((const uint16_t*)tile)[mxq_sc_u16(r)];
    sc[0] = s & 15u;
    sc[1] = (s >> 4) & 15u;
    sc[2] = (s >> 8) & 15u;
}
