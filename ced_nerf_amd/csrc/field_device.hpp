// Device-side pieces shared by the fp32 (field.hip) and half-precision (field_half.hip) fused field
// kernels: per-level hash-grid constants, the trilinear gather of one level, the temporal key-frame
// split and the 9-wide time encoding.  Everything here computes in fp32 with the deterministic
// transcendentals of ced_common.hpp, whatever precision the MLPs run in.
#pragma once
#include <cstdint>

#include "ced_common.hpp"

namespace ced {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// Tile -> wave mapping of the persistent field kernels (one workgroup per CU, WAVES waves, tiles of 16 * NT samples).
// mode 0: wave w of workgroup b takes tiles b * WAVES + w, + gridDim.x * WAVES, ...
// mode 1: a round of gridDim.x * WAVES tiles is dealt in groups of four consecutive tiles (the four SIMDs of a CU) across
//         ALL workgroups before any workgroup gets a second group.  The last, partial round of a launch (a frame's
//         launches are 4-5 rounds long) then leaves every CU with about one wave per SIMD -- which runs ~2.5x faster than
//         three sharing the MFMA pipe -- instead of a third of the CUs fully loaded and the rest idle.
// mode 2 (default; needs gridDim.x % 8 == 0, else mode 1): the same dealing inside each XCD.  Workgroup b runs on XCD
//         b % 8; the eight XCDs take eight CONTIGUOUS parts of the sample stream, so a table line wanted by
//         neighbouring samples (adjacent samples of a ray, adjacent rays of a pixel tile) is fetched into ONE L2
//         instead of up to eight (round 4: +2 % f16x2, +9 % f16 on a 15 M-sample launch).
// Returns false when the workgroup has no tile at all (it leaves before staging anything).
struct TileRange { int64_t first, end, stride; };
__device__ __forceinline__ bool field_tile_range(int mode, int64_t n_tiles, int waves, int wave, TileRange &r)
{
    const int64_t grid = gridDim.x, b = blockIdx.x;
    if (mode == 2 && (grid & 7) == 0) {
        const int64_t n_groups = (n_tiles + 3) >> 2, per_xcd = grid >> 3, region = (n_groups + 7) >> 3;
        const int64_t start = (b & 7) * region;
        const int64_t g_end = start + region < n_groups ? start + region : n_groups;
        if (start + (b >> 3) >= g_end) return false;
        r.first = (start + (int64_t)(wave >> 2) * per_xcd + (b >> 3)) * 4 + (wave & 3);
        r.end = 4 * g_end < n_tiles ? 4 * g_end : n_tiles;
        r.stride = (int64_t)(waves / 4) * per_xcd * 4;
        return true;
    }
    if ((mode ? b * 4 : b * waves) >= n_tiles) return false;
    r.first = mode ? ((int64_t)(wave >> 2) * grid + b) * 4 + (wave & 3) : b * waves + wave;
    r.end = n_tiles;
    r.stride = grid * waves;
    return true;
}

// Per-level constants, pre-multiplied by the table's bytes per entry (a power of two), so the
// corner arithmetic below produces byte offsets directly: the xor-hash commutes with the shift
// ((a^b) << s == (a<<s) ^ (b<<s)) and the dense index is linear.
struct LevelConst {
    float scale;
    uint32_t sxb, syb, szb;   // per-axis multipliers in bytes: (1, p1, p2) * EB when hashed, (1, res, res^2) * EB when dense
    uint32_t offb;            // first byte of the level
    uint32_t sizeb;           // level size in bytes (dense wrap-around)
    uint32_t maskb;           // (size - 1) * EB (hashed levels: size is a power of two)
    uint32_t hashed;
};

template <bool F16, bool TEMPORAL> struct EntryBytes { static constexpr uint32_t value = (F16 ? 4u : 8u) * (TEMPORAL ? 4u : 1u); };

__device__ __forceinline__ LevelConst make_level(float scale, uint32_t res, uint32_t offset, uint32_t size, uint32_t hashed,
                                                 uint32_t eb)
{
    LevelConst L;
    L.scale = scale;
    L.sxb = eb;
    L.syb = (hashed ? 2654435761u : res) * eb;
    L.szb = (hashed ? 805459861u : res * res) * eb;
    L.offb = offset * eb;
    L.sizeb = size * eb;
    L.maskb = (size - 1u) * eb;
    L.hashed = hashed;
    return L;
}

// The kernels keep the 16 levels' constants in LDS as 8 words each and fetch them where they are used
// (two ds_read_b128 per level per wave tile) instead of holding 32 registers across the whole tile.
__device__ __forceinline__ void store_level(uint32_t *lt, const LevelConst &L)
{
    lt[0] = __float_as_uint(L.scale);
    lt[1] = L.sxb;
    lt[2] = L.syb;
    lt[3] = L.szb;
    lt[4] = L.offb;
    lt[5] = L.sizeb;
    lt[6] = L.maskb;
    lt[7] = L.hashed;
}
__device__ __forceinline__ LevelConst load_level(const uint32_t *lt)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 a = *reinterpret_cast<const u4 *>(lt), b = *reinterpret_cast<const u4 *>(lt + 4);
    LevelConst L;
    L.scale = __uint_as_float(a[0]);
    L.sxb = a[1];
    L.syb = a[2];
    L.szb = a[3];
    L.offb = b[0];
    L.sizeb = b[1];
    L.maskb = b[2];
    L.hashed = b[3];
    return L;
}

// Trilinear gather of one level for one point (hash_encoder_half.py:112-161; temporal variant
// hash_encoder_inter.py:148-197).  x already clamped to [0,1].  MODE: 0 = this lane's level may be
// dense or hashed (both index forms computed, selected per lane), 1 = dense, 2 = hashed, 3 = dense with the x-corner
// pairs fetched by one load each (non-temporal tables, levels followed by another level).
template <bool F16, bool TEMPORAL, int MODE>
__device__ __forceinline__ void hash_level(const LevelConst &L, const void *__restrict__ table, const float (&x)[3],
                                           int k_lo, float t_frac, float &f0, float &f1)
{
    uint32_t g[3];
    float fr[3], om[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // p >= 0.5: the conversion truncates = floor, and p - floor(p) is exact, which is what v_fract_f32 returns
        const float p = x[a] * L.scale + 0.5f;
#ifdef CED_AB_NO_FRACT
        const float fl = __builtin_floorf(p);
        g[a] = (uint32_t)fl;
        fr[a] = p - fl;
#else
        g[a] = (uint32_t)p;
        fr[a] = __builtin_amdgcn_fractf(p);
#endif
        om[a] = 1.0f - fr[a];
    }
    // byte strides: x is the entry size (a power of two: a shift); dense levels have res^2 * EB < 2^24, so
    // their products are single full-rate 24-bit multiplies; hashed levels need the 32-bit wrap-around product
    constexpr uint32_t EB = EntryBytes<F16, TEMPORAL>::value;
    const uint32_t x0 = g[0] * EB;
    const uint32_t y0 = (MODE == 1 || MODE == 3) ? __umul24(g[1], L.syb) : g[1] * L.syb;
    const uint32_t z0 = (MODE == 1 || MODE == 3) ? __umul24(g[2], L.szb) : g[2] * L.szb;
    const uint32_t xs[2] = { x0, x0 + EB };
    const uint32_t ys[2] = { y0, y0 + L.syb };
    const uint32_t zs[2] = { z0, z0 + L.szb };
    const bool hashed = L.hashed != 0;
    // y/z combinations are shared by the two x corners
    uint32_t yz_x[4], yz_a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if constexpr (MODE != 1 && MODE != 3) yz_x[q] = ys[q & 1] ^ zs[q >> 1];
        if constexpr (MODE != 2) yz_a[q] = ys[q & 1] + zs[q >> 1];
    }
    const float wxy[4] = { om[0] * om[1], fr[0] * om[1], om[0] * fr[1], fr[0] * fr[1] };   // index cx + 2*cy
    uint32_t off[8];
    float w[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
        uint32_t hb = 0, db = 0;
        if constexpr (MODE != 1 && MODE != 3) hb = (xs[cx] ^ yz_x[cy + 2 * cz]) & L.maskb;
        if constexpr (MODE != 2) {
            // idx % size with idx < 2 * size: the unsigned difference wraps to a huge value when idx < size
            const uint32_t dx = xs[cx] + yz_a[cy + 2 * cz];
            const uint32_t dw = dx - L.sizeb;
            db = dx < dw ? dx : dw;
        }
        const uint32_t idxb = (MODE == 1 || MODE == 3) ? db : (MODE == 2) ? hb : (hashed ? hb : db);
#ifdef CED_AB_GATHER_WINDOW            // diagnostic builds (tools/ab_build.sh): the gathers of a level confined to a byte window
        off[c] = L.offb + (idxb & (uint32_t)(CED_AB_GATHER_WINDOW));
#else
        off[c] = L.offb + idxb;
#endif
        w[c] = wxy[cx + 2 * cy] * (cz ? fr[2] : om[2]);
    }
    const char *tb = reinterpret_cast<const char *>(table);
    // both features of a corner travel as one register pair: the interpolation is 8 packed FMAs
    // (v_pk_fma_f32 is the same IEEE fma per component as two scalar ones)
    f2 v[8];
    if constexpr (!TEMPORAL) {
        if constexpr (MODE == 3) {
            // dense level (not the table's last: field.hip): the two x corners of a (y, z) corner are adjacent entries -- ONE
            // 16-byte (fp16 table: 8-byte) load instead of two on the same cache line.  The one exception is the reference's
            // `% size` acting between the two (the +x corner is the level's entry 0): that lane re-reads its second corner.
            constexpr uint32_t EBp = EntryBytes<F16, TEMPORAL>::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (!F16) {
                    const f4 pr = *reinterpret_cast<const f4 *>(tb + off[2 * q]);
                    v[2 * q] = f2{ pr[0], pr[1] };
                    v[2 * q + 1] = f2{ pr[2], pr[3] };
                    if (off[2 * q + 1] != off[2 * q] + EBp) v[2 * q + 1] = *reinterpret_cast<const f2 *>(tb + off[2 * q + 1]);
                } else {
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                    const u2v u = *reinterpret_cast<const u2v *>(tb + off[2 * q]);
                    uint32_t u1 = u[1];
                    if (off[2 * q + 1] != off[2 * q] + EBp) u1 = *reinterpret_cast<const uint32_t *>(tb + off[2 * q + 1]);
                    v[2 * q] = f2{ half_bits_to_float((uint16_t)(u[0] & 0xffffu)), half_bits_to_float((uint16_t)(u[0] >> 16)) };
                    v[2 * q + 1] = f2{ half_bits_to_float((uint16_t)(u1 & 0xffffu)), half_bits_to_float((uint16_t)(u1 >> 16)) };
                }
            }
        } else if constexpr (!F16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const f2 *>(tb + off[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t u = *reinterpret_cast<const uint32_t *>(tb + off[c]);
                v[c] = f2{ half_bits_to_float((uint16_t)(u & 0xffffu)), half_bits_to_float((uint16_t)(u >> 16)) };
            }
        }
    } else {
        const float omt = 1.0f - t_frac;
        const f2 omt2 = { omt, omt }, tf2 = { t_frac, t_frac };
        const uint32_t kb = (uint32_t)k_lo * (F16 ? 4u : 8u);      // byte offset of key-frame k_lo inside the entry
        if constexpr (!F16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // key-frames k and k + 1 of an entry are adjacent: ONE 16-byte load (rounds 1-4: two 8-byte loads)
                const f4 pr = *reinterpret_cast<const f4 *>(tb + off[c] + kb);
                const f2 lo = { pr[0], pr[1] }, hi = { pr[2], pr[3] };
                v[c] = lo * omt2 + hi * tf2;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                typedef uint32_t u2t __attribute__((ext_vector_type(2)));
                const u2t pr = *reinterpret_cast<const u2t *>(tb + off[c] + kb);
                const uint32_t lo = pr[0], hi = pr[1];
                const f2 a = { half_bits_to_float((uint16_t)(lo & 0xffffu)), half_bits_to_float((uint16_t)(lo >> 16)) };
                const f2 b = { half_bits_to_float((uint16_t)(hi & 0xffffu)), half_bits_to_float((uint16_t)(hi >> 16)) };
                v[c] = a * omt2 + b * tf2;
            }
        }
    }
    f2 acc = { 0.0f, 0.0f };
#pragma unroll
    for (int c = 0; c < 8; ++c) acc = __builtin_elementwise_fma(f2{ w[c], w[c] }, v[c], acc);
    f0 = acc[0];
    f1 = acc[1];
}

__device__ __forceinline__ void temporal_keyframe(float tq, int &k_lo, float &t_frac)
{
    float ts = tq * 3.0f;
    float fl = __builtin_floorf(ts);
    t_frac = ts - fl;
    fl = __builtin_fminf(fl, 2.0f);
    k_lo = (int)fl;
}

// feature idx (0..8, >8 -> 0) of the 9-wide time encoding (cednerf/encoder.py:6-44 / :46-90)
__device__ __forceinline__ float time_feature(int idx, int time_mode, float t, float mn)
{
    const float HALF_PI = 1.57079637050628662f;
    if (idx == 0) return t;
    if (idx > 8) return 0.0f;
    int k, ph;
    if (time_mode == 1) { k = (idx - 1) & 3; ph = (idx - 1) >> 2; }
    else { k = (idx - 1) >> 1; ph = (idx - 1) & 1; }
    float xb = t * (float)(1 << k);
    float arg = ph ? (xb + HALF_PI) : xb;
    float s = det_sinf(arg);
    if (time_mode == 2) {
        float att = det_expf(-1.0f * (mn * (float)(k * (1 << k))));
        s = s * att;
    }
    return s;
}

}  // namespace ced
