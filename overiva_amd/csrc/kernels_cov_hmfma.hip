// Weighted spatial covariance pass for 10..16 channels and MANY sources (9..16: BASELINE configs[4] is the determined
// 16 x 16 case) with the SOURCES on the matrix cores.
//
//   V_k[f] = sum_t rinv[t,k] * x_{t,f} x_{t,f}^H          reference overiva.py:179, all K sources in one pass over X
//
// Per (bin, frame) the Hermitian half of x x^H is M^2 real numbers, and the weighted sums of ALL sources are one small GEMM
// per bin:   V[k][e] = sum_t w[k][t] * H[t][e]        k: 16 sources, e: the packed entries, t: frames
// -- per 4 frames one v_mfma_f32_16x16x4_f32 per group of 16 entries: A = the weights (16 sources x 4 frames), B = 16
// entries of H for those 4 frames, D = 16 sources x 16 entries -- for all 16 sources together, where the planar
// matrix-core kernel (kernels_cov_mfma.hip: rank-1 updates of full real 16 x 16 tiles, 3 instructions per 4 frames and
// SOURCE) issues 48 and the vector-ALU kernel (kernels_cov_half16.hip) spends one packed FMA per entry and source.
// The 16 entries of a group are the CYCLIC diagonal c of the matrix: lane n forms the product of channel n with channel
// (n + c) mod M -- c = 0: |x_n|^2; c = 1 .. M/2: real and imaginary part (two groups; at c = M/2 the lanes n >= M/2 repeat
// the others and are dropped) -- so that every lane runs the same two instructions per group (a multiply and an FMA, no
// select), which is what fits beside a matrix instruction: the pipe takes one every 32 cycles and hides about five issue
// slots (MI355X_MICROARCH.md).  M + 1 matrix instructions per bin and 4 frames (17 at 16 channels).  The fp32 matrix
// instruction is an exact fmaf chain in frame order, so the arithmetic class is that of the vector-ALU kernel: float32
// products, float32 chains of T / (8 nsplit) frames, float64 sums across chains, waves and splits.
//
// Geometry: a workgroup = ONE bin x 4 waves x 2 accumulator sets = 8 frame phases.  A wave's frames are t_begin + wave + 4 n;
// a stage holds n = 8 i .. 8 i + 7, the first four feeding accumulator set 0 and the last four set 1: two independent
// float32 chains per wave, so that the chain bound of the arithmetic class needs HALF the frame splits of a kernel with one
// chain per wave -- half the float64 partials written here and read by the update kernel (2048 x 4000 x 16 / 16: 4 splits,
// 268 MB, instead of 8 and 537 MB).  One global_load_lds per wave and stage moves 8 frames x 128 bytes into a 4-stage ring;
// the 128 weights of a stage (8 frames x 16 sources) ride the same ring by two 4-byte DMAs whose lane l lands exactly where
// lane l reads its A operand.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "oiva_device.h"
#include "cov_arith.h"

namespace oiva {
namespace {

constexpr int kHmStages = 4;
constexpr int kHmFrames = 4;                            // frames per stage of a wave = the contraction of one MFMA
constexpr int kHmSlot = 256;                            // bytes of two frames (one per accumulator set) x (<= 16) channels, 128 each
constexpr int kHmX = kHmFrames * kHmSlot;               // 1 KB of X per stage per wave
constexpr int kHmStage = kHmX + 512;                    // + 2 x 64 weights
constexpr int kHmLdsStride = kBlock + 1;
constexpr int kHmWeightStride = 16;                     // row stride of the weight table (launch_cov_weights)
constexpr int kWtHalf = 4 * kHmFrames * kHmWeightStride * 4;      // bytes of the weight table between a frame and the one 16 later
static_assert(kWtHalf <= kHmX + 256, "the LDS base shifted by it stays inside the stage");

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// The lane's operands of one accumulator set of one stage: its own channel n and the channels (n + c) mod M, c = 1 .. 8 (nine
// 8-byte LDS reads at per-lane addresses) and its weight -- for the first set with the counted wait for the stage's three
// DMAs in front.  asm: hipcc drains the whole DMA queue in front of any LDS read it can see.  The reads are ISSUED by one
// statement and WAITED for by another, so that the second set's reads fly while the first set's matrix instructions issue.
// XOFF / WOFF: the stage and set as the instruction's immediate offset (the loop is unrolled over the ring, round 5: the
// per-stage address additions were 4 of the 93 vector instructions beside the 34 matrix instructions of a stage).
struct HmOps {
    float2 row;
    float2 x[8];
};
template <bool FIRST, bool M16, int XOFF, int WOFF>
__device__ __forceinline__ void hm_read_issue(const unsigned (&ax)[9], unsigned a_w, HmOps& o, float& w) {
    if constexpr (FIRST) {
        asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b32 %0, %1 offset:%3" : "=&v"(w) : "v"(a_w), "n"(3 * (kHmStages - 1)), "n"(WOFF) : "memory");
    } else {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=&v"(w) : "v"(a_w), "n"(WOFF) : "memory");
    }
    if constexpr (M16) {
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(o.row) : "v"(ax[0]), "n"(XOFF) : "memory");
        return;
    }
    asm volatile(
        "ds_read_b64 %0, %9 offset:%18\n\t"
        "ds_read_b64 %1, %10 offset:%18\n\t"
        "ds_read_b64 %2, %11 offset:%18\n\t"
        "ds_read_b64 %3, %12 offset:%18\n\t"
        "ds_read_b64 %4, %13 offset:%18\n\t"
        "ds_read_b64 %5, %14 offset:%18\n\t"
        "ds_read_b64 %6, %15 offset:%18\n\t"
        "ds_read_b64 %7, %16 offset:%18\n\t"
        "ds_read_b64 %8, %17 offset:%18"
        : "=&v"(o.row), "=&v"(o.x[0]), "=&v"(o.x[1]), "=&v"(o.x[2]), "=&v"(o.x[3]), "=&v"(o.x[4]), "=&v"(o.x[5]), "=&v"(o.x[6]), "=&v"(o.x[7])
        : "v"(ax[0]), "v"(ax[1]), "v"(ax[2]), "v"(ax[3]), "v"(ax[4]), "v"(ax[5]), "v"(ax[6]), "v"(ax[7]), "v"(ax[8]), "n"(XOFF)
        : "memory");
}
// (the wait names no register -- tied operands of struct members are not supported -- so a scheduling barrier behind it keeps
//  every consumer of the destinations below it)
__device__ __forceinline__ void hm_read_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int I>
using hm_ic = std::integral_constant<int, I>;

// M16: exactly 16 channels -- channel (n + c) mod 16 of the lane's frame sits c lanes further in its 16-lane row, so the
// partner of every product is a DPP row rotation of the lane's own operand (a modifier of the multiply / FMA itself): one
// 8-byte LDS read per lane, set and stage instead of nine (which, four frames hitting the same banks, kept the LDS pipe of
// the CU busier than the matrix pipes).  Fewer channels: the partners are gathered from LDS.
//   With 16 channels the distance c = 8 pairs every channel with its opposite twice: its real and imaginary parts share ONE
//   group (lanes 0..7 the real part of (n, n + 8), lanes 8..15 the imaginary part of (n - 8, n), by the bank mask of the DPP
//   instructions) -- 16 matrix instructions per 4 frames, the 256 real numbers of the Hermitian half exactly, instead of 17.
// BUF: the DMAs in buffer form -- the resource descriptor (base, bytes left) steps through the frames on the SCALAR unit, the
// lane's offset is a constant, frames past the end read as zero by the descriptor's range check: no vector instruction for
// addresses (the flat form spent 21 per stage on them, three at a quarter of the rate; on this chip vector and matrix
// instructions share the ALUs and their times add).  Needs a split's frames + 35 within 4 GB of X; else the flat form.
template <bool M16, bool BUF, bool PK>
__global__ __launch_bounds__(kBlock, 2) void cov_hmfma_kernel(const float2* __restrict__ X, const float* __restrict__ Wt,
                                                              double* __restrict__ Vpart, int T, int F, int M, int Mv, int K, int tc, int part32) {
    constexpr int NG = M16 ? 16 : 17;
    constexpr int kRingBytes = kWaves * kHmStages * kHmStage;
    constexpr int kScratchBytes = 16 * 8 * kBlock;       // reduction scratch: 4 groups x 2 sets of 16-byte vectors per thread
    __shared__ float4 ring[(kRingBytes > kScratchBytes ? kRingBytes : kScratchBytes) / 16 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;                            // frame of the stage (contraction index)
    const int n = lane & 15;                            // A: source | B: row channel i
    const int f0 = blockIdx.x;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nstages = (t_end - t_begin + 8 * kHmFrames - 1) / (8 * kHmFrames);

    // [accumulator set][group]: group 0 = |x_n|^2, 2 c - 1 / 2 c = real / imaginary part of x_n conj(x_(n + c) mod M), c = 1 .. 8
    // (M16: c = 1 .. 7, group 15 = both parts of c = 8); lane (q, n) holds sources 4 q + r of its entry
    f32x4 acc[2][NG];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < NG; ++j) acc[h][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int MH = M / 2;

    // ---- DMA side: lane l moves 16-byte piece l & 7 of frame slot l >> 4, half (l & 15) >> 3: the half-0 frame of slot q is
    //      the wave's frame n = 8 i + q, the half-1 frame n = 8 i + 4 + q (16 frames later).  The weights of the same frames:
    //      lane l moves Wt[frame of slot l >> 4][source l & 15], once per half (frames past the split: the zeroed row T).
    char* wring = reinterpret_cast<char*>(ring) + wave * (kHmStages * kHmStage);       // wave-uniform
    // (PK: the frames of a stage lie set-major, 128 bytes each -- lane l moves piece l & 7 of frame (l >> 3) & 3 of set l >> 5 --, so
    //  that the four frames of an operand read fall into two bank groups instead of one: see stage_pk below)
    const int half = PK ? lane >> 5 : (lane >> 3) & 1;
    const int qd = PK ? (lane >> 3) & 3 : q;              // frame slot of the lane's DMA piece
    const unsigned piece_off = (unsigned)min(lane & 7, M / 2 - 1) * 16u;
    const char* xbytes = reinterpret_cast<const char*>(X);
    const size_t row_bytes = (size_t)F * M * 8;
    const char* run0 = xbytes + (size_t)f0 * M * 8 + piece_off;
    // (buffer form) the workgroup's bin at the first frame of its split is offset 0 of stage 0's descriptor; a stage is 32 frames
    const unsigned row32 = (unsigned)row_bytes;            // (the launch checks 36 rows < 4 GB)
    const unsigned col_off = (unsigned)f0 * (unsigned)M * 8u;
    const unsigned xvoff = (unsigned)(wave + 4 * qd + 16 * half) * row32 + piece_off;
    const unsigned wvoff0 = (unsigned)((wave + 4 * q) * kHmWeightStride + n) * 4u;
    auto issue = [&](int i, auto sc) {
        constexpr int s = decltype(sc)::value;
        char* dst = wring + s * kHmStage;
        if constexpr (BUF) {
            // (all of this on the scalar unit: 32-bit compares only -- there is no scalar 64-bit compare)
            const int tx = t_begin + 8 * kHmFrames * i, fl = T - tx;                                  // frames left
            // bytes of X from the descriptor's base on, as far as this stage's 32 frames reach (min / max only: a select would
            // leave the scalar unit and the descriptor would be rebuilt per lane)
            //   rec = max(min(max(fl, 0), 32) * row32, col_off) - col_off
            // (asm: hipcc picks v_med3_i32 and a clamped vector subtraction for this, and a descriptor word that lives in a vector
            //  register is applied lane by lane in a loop)
            unsigned rec;
            asm("s_max_i32 %0, %1, 0\n\t"
                "s_min_i32 %0, %0, %4\n\t"
                "s_mul_i32 %0, %0, %2\n\t"
                "s_max_u32 %0, %0, %3\n\t"
                "s_sub_u32 %0, %0, %3"
                : "=&s"(rec)
                : "s"(fl), "s"(row32), "s"(col_off), "n"(8 * kHmFrames)
                : "scc");
            const char* xb = xbytes + ((size_t)(unsigned)min(tx, T) * row32 + col_off);
            const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)rec, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lvoid_t*)dst, 16, (int)xvoff, 0, 0, 0);
            const int tw = min(t_begin + 8 * kHmFrames * i, T);
            const auto wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt + (size_t)tw * kHmWeightStride), 0, (T - tw) * kHmWeightStride * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lvoid_t*)(dst + kHmX), 4, (int)wvoff0, 0, 0, 0);
            // (the second half's weights 16 frames = 1 KB further in the table: as the instruction's immediate offset, which moves
            //  the LDS address by as much -- so the LDS base is handed over 1 KB short; one address register less, which is what
            //  keeps the PK form at three waves per SIMD)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lvoid_t*)(dst + kHmX + 256 - kWtHalf), 4, (int)wvoff0, 0, kWtHalf, 0);
        } else {
            const int t0 = t_begin + wave + 4 * (2 * kHmFrames * i + q), t1 = t0 + 4 * kHmFrames;
            const int tx = t_begin + wave + 4 * (2 * kHmFrames * i + qd) + (half ? 4 * kHmFrames : 0);
            const int tcl = min(i < nstages ? tx : T - 1, T - 1);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(run0 + (size_t)tcl * row_bytes), (lvoid_t*)dst, 16, 0, 0);
            const int tw0 = (i < nstages && t0 < t_end) ? t0 : T, tw1 = (i < nstages && t1 < t_end) ? t1 : T;
            __builtin_amdgcn_global_load_lds((gvoid_t*)(Wt + (size_t)tw0 * kHmWeightStride + n), (lvoid_t*)(dst + kHmX), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(Wt + (size_t)tw1 * kHmWeightStride + n), (lvoid_t*)(dst + kHmX + 256), 4, 0, 0);
        }
    };

    // ---- operand addresses of this lane inside a frame slot of set 0 of stage 0 (set 1 at + 128 bytes, stage s at + s stages):
    //      its own channel, then the channels (n + c) mod M; lanes n >= M (fewer than 16 channels) read channel 0 and produce
    //      entries that are dropped
    const unsigned lbase = (unsigned)(uintptr_t)wring + (unsigned)(q * (PK ? kHmSlot / 2 : kHmSlot));
    const int nn = n < M ? n : 0;
    unsigned ax[9];
    ax[0] = lbase + 8u * (unsigned)nn;
#pragma unroll
    for (int c = 1; c <= 8; ++c) ax[c] = lbase + 8u * (unsigned)((nn + c) % M);
    const unsigned a_w = (unsigned)(uintptr_t)wring + 4u * (unsigned)lane;
    constexpr int setoff = 128;

    // the groups of one accumulator set: a product (multiply, FMA onto it) and one MFMA with the stage's weights each
    auto groups = [&](const HmOps& o, float w, f32x4 (&a)[NG]) {
        float both8;
        if constexpr (M16) {
            // c = 8: lanes 0..7 of a row  Re x_n conj x_(n + 8) = fma(xi_n, xi_m, xr_n xr_m),
            //        lanes 8..15 (own channel m = n' + 8, partner n')  Im x_n' conj x_m = fma(xi_n', xr_m, -(xr_n' xi_m))
            // -- the same operations in the same order as the two groups this one replaces.  (s_nop: the matrix instruction
            // that follows reads the register the last instruction of the block writes, see below.)
            asm("v_mul_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                "v_mul_f32_dpp %0, -%1, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                "v_fmac_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                "v_fmac_f32_dpp %0, %2, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                "s_nop 1"
                : "=&v"(both8)
                : "v"(o.row.x), "v"(o.row.y));
        }
        a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, fmaf(o.row.y, o.row.y, o.row.x * o.row.x), a[0], 0, 0, 0);
        if constexpr (M16) a[15] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, both8, a[15], 0, 0, 0);
        static_for<M16 ? 7 : 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value + 1;
            if constexpr (M16) {
                // the partner channel m = (n + c) mod 16 is the lane's own operand rotated by c lanes inside its 16-lane row
                // (row_ror:(16 - c): lane n receives lane n + c), as a DPP modifier of the multiply / FMA itself: four vector
                // instructions per pair of matrix instructions -- on this chip the fp32 matrix instruction runs on the same
                // ALUs as the vector instructions (measured: their times ADD), so every vector instruction here counts.
                //   re = fma(row.y, xi_m, row.x * xr_m)          im = fma(row.y, xr_m, -(row.x * xi_m))
                // (Order: the two chains interleaved -- mul re, fmac re, mul im, fmac im measured 4-7 % slower on the whole kernel --
                //  and `re`, which the first matrix instruction reads, complete one vector instruction before the block ends: a
                //  matrix instruction reading a register written by the LAST instruction of an asm block gets the OLD value, since
                //  hipcc does not see the write and inserts no wait state; measured: two wait states (`s_nop 1`) suffice, a
                //  64-lane vector instruction is four.)
                float re, im;
                asm("v_mul_f32_dpp %0, %2, %2 row_ror:%4 row_mask:0xf bank_mask:0xf\n\t"
                    "v_mul_f32_dpp %1, -%3, %2 row_ror:%4 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %0, %3, %3 row_ror:%4 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f32_dpp %1, %2, %3 row_ror:%4 row_mask:0xf bank_mask:0xf"
                    : "=&v"(re), "=&v"(im)
                    : "v"(o.row.x), "v"(o.row.y), "n"(16 - c));
                a[2 * c - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, re, a[2 * c - 1], 0, 0, 0);
                a[2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, im, a[2 * c], 0, 0, 0);
            } else if (c <= MH) {                             // (wave-uniform)
                const float2 x = o.x[c - 1];
                a[2 * c - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, fmaf(o.row.y, x.y, o.row.x * x.x), a[2 * c - 1], 0, 0, 0);     // Re x_n conj x_m
                a[2 * c] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, fmaf(o.row.y, x.x, -(o.row.x * x.y)), a[2 * c], 0, 0, 0);         // Im x_n conj x_m
            }
        });
    };
    // one stage: both sets; the second set's operands are read while the first set's matrix instructions issue
    auto stage = [&](auto sc) {
        constexpr int so = decltype(sc)::value * kHmStage;
        float w0, w1;
        HmOps o0, o1;
        hm_read_issue<true, M16, so, so + kHmX>(ax, a_w, o0, w0);
        hm_read_wait();
        hm_read_issue<false, M16, so + setoff, so + kHmX + 256>(ax, a_w, o1, w1);
        groups(o0, w0, acc[0]);
        hm_read_wait();
        groups(o1, w1, acc[1]);
    };

    // PK (16 channels): the partners of the products come from LDS -- one ds_read_b64 per pair distance at the lane's rotated
    // address -- and a product pair (re, im) is TWO packed instructions (v_pk_mul_f32 / v_pk_fma_f32 with op_sel: cov_arith.h)
    // instead of four DPP ones: 20 vector instructions per accumulator set (2 for the diagonal, 4 for the shared group of distance
    // 8, 2 x 7) instead of 34 beside its 16 matrix instructions.  Same operations in the same order: same bits.  The frames of a
    // stage lie set-major, 128 bytes each: lane (q, n) reads byte 128 q + 8 ((n + c) mod 16) of its set -- 32 different 8-byte
    // bank pairs, each hit by two of the four frames, the minimum of a 64-lane 8-byte read (frame slots of 256 bytes put all four
    // frames on the same banks: four passes).  Reads in batches of at most six, issued two batches ahead of their use, counted
    // lgkmcnt waits; the products run one group ahead of the matrix instructions that read them (a matrix instruction must not
    // directly follow the asm block that writes its operand), pinned by scheduling barriers.
    constexpr int kPkSet = 512;
    auto rd64 = [&](unsigned addr, auto off) {
        v2f r;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value) : "memory");
        return r;
    };
    auto rd32 = [&](unsigned addr, auto off) {
        float r;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value) : "memory");
        return r;
    };
    // (re, im) of row conj(part): (row.x part.x, -(row.x part.y)), then += (row.y part.y, row.y part.x) -- qk_mul_lo_negim and
    // qk_fma_hi_swap of cov_arith.h as ONE statement (hipcc puts a wait state between two asm statements that share a register)
    auto prod = [&](v2f row, v2f part) {
        v2f p;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
                     : "=&v"(p)
                     : "v"(row), "v"(part));
        return p;
    };
    auto mm = [&](f32x4& a, float w, float b) { a = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a, 0, 0, 0); };
    // diagonal + distance 8 (its two parts in one group, by DPP) + distances 1 .. 4 of one set; pa: the partners 1 .. 4
    auto set_first = [&](v2f row, float w, const v2f (&pa)[4], f32x4 (&a)[NG]) {
        float both8;
        asm volatile("v_mul_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_mul_f32_dpp %0, -%1, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "v_fmac_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_fmac_f32_dpp %0, %2, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
                     : "=&v"(both8)
                     : "v"(row.x), "v"(row.y));
        const v2f p1 = prod(row, pa[0]);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[0], w, fmaf(row.y, row.y, row.x * row.x));
        mm(a[15], w, both8);
        const v2f p2 = prod(row, pa[1]);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[1], w, p1.x);
        mm(a[2], w, p1.y);
        const v2f p3 = prod(row, pa[2]);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[3], w, p2.x);
        mm(a[4], w, p2.y);
        const v2f p4 = prod(row, pa[3]);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[5], w, p3.x);
        mm(a[6], w, p3.y);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[7], w, p4.x);
        mm(a[8], w, p4.y);
        __builtin_amdgcn_sched_barrier(0);
    };
    // distances 5 .. 7
    auto set_second = [&](v2f row, float w, const v2f (&pb)[3], f32x4 (&a)[NG]) {
        const v2f p5 = prod(row, pb[0]);
        const v2f p6 = prod(row, pb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[9], w, p5.x);
        mm(a[10], w, p5.y);
        const v2f p7 = prod(row, pb[2]);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[11], w, p6.x);
        mm(a[12], w, p6.y);
        __builtin_amdgcn_sched_barrier(0);
        mm(a[13], w, p7.x);
        mm(a[14], w, p7.y);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_pk = [&](auto sc) {
        constexpr int so = decltype(sc)::value * kHmStage;
        using O0 = hm_ic<so>;
        using O1 = hm_ic<so + kPkSet>;
        // batch A: weight, own sample and partners 1 .. 4 of set 0, behind the counted wait for the stage's three DMAs
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (kHmStages - 1)) : "memory");
        const float w0 = rd32(a_w, hm_ic<so + kHmX>{});
        const v2f r0 = rd64(ax[0], O0{});
        v2f pa[4], pb[3];
        pa[0] = rd64(ax[1], O0{});
        pa[1] = rd64(ax[2], O0{});
        pa[2] = rd64(ax[3], O0{});
        pa[3] = rd64(ax[4], O0{});
        // batch B: partners 5 .. 7 of set 0, weight and own sample of set 1
        pb[0] = rd64(ax[5], O0{});
        pb[1] = rd64(ax[6], O0{});
        pb[2] = rd64(ax[7], O0{});
        const float w1 = rd32(a_w, hm_ic<so + kHmX + 256>{});
        const v2f r1 = rd64(ax[0], O1{});
        asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");       // batch A has arrived (LDS reads return in order)
        __builtin_amdgcn_sched_barrier(0);
        set_first(r0, w0, pa, acc[0]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // batch B
        __builtin_amdgcn_sched_barrier(0);
        // batch C: partners 1 .. 4 of set 1 (into the registers set 0 has finished with)
        v2f qa[4], qb[3];
        qa[0] = rd64(ax[1], O1{});
        qa[1] = rd64(ax[2], O1{});
        qa[2] = rd64(ax[3], O1{});
        qa[3] = rd64(ax[4], O1{});
        __builtin_amdgcn_sched_barrier(0);
        set_second(r0, w0, pb, acc[0]);
        // batch D: partners 5 .. 7 of set 1
        qb[0] = rd64(ax[5], O1{});
        qb[1] = rd64(ax[6], O1{});
        qb[2] = rd64(ax[7], O1{});
        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");       // batch C
        __builtin_amdgcn_sched_barrier(0);
        set_first(r1, w1, qa, acc[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // batch D
        __builtin_amdgcn_sched_barrier(0);
        set_second(r1, w1, qb, acc[1]);
    };
    auto run_stage = [&](auto sc) {
        if constexpr (PK)
            stage_pk(sc);
        else
            stage(sc);
    };

    // the ring of four stages unrolled: stage i sits in slot i mod 4, its successor i + 3 is requested before it is consumed
    // (every stage of the loop requests exactly three DMAs -- the counted wait of the first read depends on it; requests past
    //  the split's last stage land in slots nobody reads)
    issue(0, hm_ic<0>{});
    issue(1, hm_ic<1>{});
    issue(2, hm_ic<2>{});
    int i = 0;
    for (; i + 4 <= nstages; i += 4) {
        issue(i + 3, hm_ic<3>{});
        run_stage(hm_ic<0>{});
        issue(i + 4, hm_ic<0>{});
        run_stage(hm_ic<1>{});
        issue(i + 5, hm_ic<1>{});
        run_stage(hm_ic<2>{});
        issue(i + 6, hm_ic<2>{});
        run_stage(hm_ic<3>{});
    }
    if (i < nstages) {
        issue(i + 3, hm_ic<3>{});
        run_stage(hm_ic<0>{});
        ++i;
    }
    if (i < nstages) {
        issue(i + 3, hm_ic<0>{});
        run_stage(hm_ic<1>{});
        ++i;
    }
    if (i < nstages) {
        issue(i + 3, hm_ic<1>{});
        run_stage(hm_ic<2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the DMA queue before the ring becomes reduction scratch

    // ---- the eight chains (4 waves x 2 sets) added in float64, fixed order; accumulator r of group grp of lane (q, n) is
    //      source 4 q + r, entry: grp 0 the diagonal n; grp 2 c - 1 / 2 c the real / imaginary part of the pair
    //      (n, (n + c) mod M), stored under its ordered form (i < j): the imaginary part changes sign when the pair wraps
    //      (M16, grp 15: lanes 0..7 the real part of (n, n + 8), lanes 8..15 the imaginary part of (n - 8, n)).
    //      A round = 4 groups; wave w adds the eight values of accumulator r = w of each, so that group, pair distance and
    //      re / im are compile-time and only (source row, channel) come from the lane.
    // (round 6) A round = 4 groups; wave w adds GROUP w of the round -- all four accumulator registers (sources 4 q .. 4 q + 3) of
    // it, so that the accumulators cross LDS as whole 16-byte vectors: 8 ds_write_b128 + 8 ds_read_b128 per lane and round where
    // the round-4/5 form (wave w adds register w of every group) moved them one float at a time, 32 + 32 -- the same eight values
    // in the same order into every sum, the same bits; the pass at four splits 633 -> 6xx us.  The group, and with it pair distance
    // and re / im, is wave-uniform now (scalar registers), no longer compile-time.
    f32x4* lds4 = reinterpret_cast<f32x4*>(ring);          // [group of the round][set][thread]
    const int NA = Mv * Mv;
    // (round 6) part32: the block leaves as float32 -- each value the float64 sum of 8 float32 chains of <= 128 frames, rounded ONCE
    // (6e-8 relative, a tenth of what its chains carry; the update adds the splits' blocks in float64 as before): half the bytes
    // written here and read by the per-bin update, which at 16 x 16 is bound by exactly those bytes (277 MB = 46 us of 66)
    const size_t vbase = (((size_t)blockIdx.y * F + f0) * K + 4 * q) * NA;
    double* vout = Vpart + vbase;
    float* vout32 = reinterpret_cast<float*>(Vpart) + vbase;
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += 4) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (g0 + v < NG) lds4[(v * 2 + h) * kBlock + tid] = acc[h][g0 + v];
        __syncthreads();
        const int grp = g0 + wave;                  // (wave-uniform)
        if (grp >= NG) continue;
        double s[4] = {0., 0., 0., 0.};
#pragma unroll
        for (int w = 0; w < kWaves; ++w)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 t = lds4[(wave * 2 + h) * kBlock + w * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] += (double)t[r];
            }
        if (n >= M) continue;
        const int c = (grp + 1) >> 1, im = (grp + 1) & 1;            // grp 2c-1: re, 2c: im   (grp 0: the diagonal)
        int pos;
        bool neg = false;
        if (grp == 0) {
            if (n >= Mv) continue;
            pos = n;
        } else if (M16 && grp == 15) {
            const int i8 = n & 7, j8 = i8 + 8;
            if (j8 >= Mv) continue;
            pos = herm_pair_index(Mv, i8, j8) + (n >> 3);
        } else {
            if (c > MH || (c == MH && n >= MH)) continue;             // (c = M/2: the upper half of the lanes repeats the lower)
            const int mm = n + c >= M ? n + c - M : n + c;
            const int i = n < mm ? n : mm, j = n < mm ? mm : n;
            if (j >= Mv) continue;
            pos = herm_pair_index(Mv, i, j) + im;
            neg = im && mm < n;                                       // Im(x_i conj x_j) = -Im(x_j conj x_i)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < K) {
                const double v = neg ? -s[r] : s[r];
                if (part32)                              // (uniform)
                    vout32[(size_t)r * NA + pos] = (float)v;
                else
                    vout[(size_t)r * NA + pos] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The `precise` arithmetic (float64 sums of exact float64 products) on the fp64 matrix cores, 16 channels, 9..16 sources:
// the same GEMM per bin with v_mfma_f64_16x16x4_f64 (A = the float64 weights, B = the products, formed in float64 from the
// lane's own sample and the DPP-rotated partner: two moves, two conversions, four float64 instructions per pair of matrix
// instructions).  One accumulator set (no chain to bound); the 8 frames of a stage go into it one half after the other.
// Weights: the float64 table of kernels_cov_half16.hip, (T + 1, 16) doubles, row T zero -- a lane's weight arrives as two
// 4-byte DMAs (low and high word, 256 bytes apart) and is read back by one ds_read2_b32.
// C/D layout of the float64 instruction: lane (q, n), register r = source q + 4 r (not the float32 map).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kHm64Stage = kHmX + 1024;                 // + 2 halves x (low, high) x 64 words
constexpr int kHm64Chunk = 16;                          // doubles per LDS round of the epilogue: 4 groups x 4

template <bool BUF>
__global__ __launch_bounds__(kBlock, 2) void cov_hmfma64_kernel(const float2* __restrict__ X, const double* __restrict__ Wt,
                                                                double* __restrict__ Vpart, int T, int F, int Mv, int K, int tc) {
    constexpr int M = 16, MH = 8;
    constexpr int kRingBytes = kWaves * kHmStages * kHm64Stage;
    constexpr int kScratchBytes = (int)sizeof(double) * kHm64Chunk * kHmLdsStride;
    __shared__ float4 ring[(kRingBytes > kScratchBytes ? kRingBytes : kScratchBytes) / 16 + 1];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;
    const int n = lane & 15;
    const int f0 = blockIdx.x;
    const int t_begin = blockIdx.y * tc;
    const int t_end = min(T, t_begin + tc);
    const int nstages = (t_end - t_begin + 8 * kHmFrames - 1) / (8 * kHmFrames);

    f64x4 acc[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) acc[j] = f64x4{0., 0., 0., 0.};

    char* wring = reinterpret_cast<char*>(ring) + wave * (kHmStages * kHm64Stage);       // wave-uniform
    const int half = (lane >> 3) & 1;
    const char* xbytes = reinterpret_cast<const char*>(X);
    const char* run0 = xbytes + (size_t)f0 * M * 8 + (unsigned)(lane & 7) * 16u;
    const size_t row_bytes = (size_t)F * M * 8;
    const float* wt32 = reinterpret_cast<const float*>(Wt);
    // (buffer form, as in the float32 kernel: descriptors stepped on the scalar unit, constant lane offsets, range-checked tails)
    const unsigned row32 = (unsigned)row_bytes;
    const unsigned col_off = (unsigned)f0 * (unsigned)M * 8u;
    const unsigned xvoff = (unsigned)(wave + 4 * q + 16 * half) * row32 + (unsigned)(lane & 7) * 16u;
    const unsigned wv0 = (unsigned)((wave + 4 * q) * kHmWeightStride + n) * 8u, wv1 = wv0 + 4u;
    const unsigned wv2 = wv0 + 4u * kHmFrames * kHmWeightStride * 8u, wv3 = wv2 + 4u;
    auto issue = [&](int i, auto sc) {
        constexpr int s = decltype(sc)::value;
        char* dst = wring + s * kHm64Stage;
        if constexpr (BUF) {
            const int tx = t_begin + 8 * kHmFrames * i, fl = T - tx;
            unsigned rec;
            asm("s_max_i32 %0, %1, 0\n\t"
                "s_min_i32 %0, %0, %4\n\t"
                "s_mul_i32 %0, %0, %2\n\t"
                "s_max_u32 %0, %0, %3\n\t"
                "s_sub_u32 %0, %0, %3"
                : "=&s"(rec)
                : "s"(fl), "s"(row32), "s"(col_off), "n"(8 * kHmFrames)
                : "scc");
            const int tw = min(tx, T);
            const char* xb = xbytes + ((size_t)(unsigned)tw * row32 + col_off);
            const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xb), 0, (int)rec, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lvoid_t*)dst, 16, (int)xvoff, 0, 0, 0);
            const auto wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Wt + (size_t)tw * kHmWeightStride), 0, (T - tw) * kHmWeightStride * 8, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lvoid_t*)(dst + kHmX), 4, (int)wv0, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lvoid_t*)(dst + kHmX + 256), 4, (int)wv1, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lvoid_t*)(dst + kHmX + 512), 4, (int)wv2, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lvoid_t*)(dst + kHmX + 768), 4, (int)wv3, 0, 0, 0);
        } else {
            const int t0 = t_begin + wave + 4 * (2 * kHmFrames * i + q), t1 = t0 + 4 * kHmFrames;
            const int tx = half ? t1 : t0;
            const int tcl = min(i < nstages ? tx : T - 1, T - 1);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(run0 + (size_t)tcl * row_bytes), (lvoid_t*)dst, 16, 0, 0);
            const int tw0 = (i < nstages && t0 < t_end) ? t0 : T, tw1 = (i < nstages && t1 < t_end) ? t1 : T;
            const float* w0 = wt32 + ((size_t)tw0 * kHmWeightStride + n) * 2;
            const float* w1 = wt32 + ((size_t)tw1 * kHmWeightStride + n) * 2;
            __builtin_amdgcn_global_load_lds((gvoid_t*)w0, (lvoid_t*)(dst + kHmX), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(w0 + 1), (lvoid_t*)(dst + kHmX + 256), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoid_t*)w1, (lvoid_t*)(dst + kHmX + 512), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoid_t*)(w1 + 1), (lvoid_t*)(dst + kHmX + 768), 4, 0, 0);
        }
    };
    const unsigned a_x = (unsigned)(uintptr_t)wring + (unsigned)(q * kHmSlot + 8 * n);
    unsigned a_ws[kHmStages];
#pragma unroll
    for (int s = 0; s < kHmStages; ++s) a_ws[s] = (unsigned)(uintptr_t)wring + (unsigned)(s * kHm64Stage + kHmX) + 4u * (unsigned)lane;

    auto groups = [&](v2f row, double w) {
        const double xr = (double)row.x, xi = (double)row.y;
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, fma(xi, xi, xr * xr), acc[0], 0, 0, 0);
        static_for<8>([&](auto cc) {
            constexpr int c = decltype(cc)::value + 1;
            // row_ror:(16 - c): lane n receives lane n + c
            const double pr = (double)dpp<0x120 + 16 - c>(row.x), pi = (double)dpp<0x120 + 16 - c>(row.y);
            acc[2 * c - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, fma(xi, pi, xr * pr), acc[2 * c - 1], 0, 0, 0);     // Re x_n conj x_m
            acc[2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(w, fma(xi, pr, -(xr * pi)), acc[2 * c], 0, 0, 0);           // Im x_n conj x_m
        });
    };
    auto stage = [&](auto sc) {
        constexpr int so = decltype(sc)::value * kHm64Stage;
        v2f r0, r1;
        double w0, w1;
        // (the 8-bit offsets of ds_read2_b32 do not reach across stages: one address register per stage for the weights)
        asm volatile("s_waitcnt vmcnt(%4)\n\t"
                     "ds_read_b64 %0, %5 offset:%7\n\t"
                     "ds_read2_b32 %2, %6 offset1:64\n\t"
                     "ds_read_b64 %1, %5 offset:%8\n\t"
                     "ds_read2_b32 %3, %6 offset0:128 offset1:192\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(w0), "=&v"(w1)
                     : "n"(5 * (kHmStages - 1)), "v"(a_x), "v"(a_ws[decltype(sc)::value]), "n"(so), "n"(so + 128)
                     : "memory");
        groups(r0, w0);
        groups(r1, w1);
    };

    issue(0, hm_ic<0>{});
    issue(1, hm_ic<1>{});
    issue(2, hm_ic<2>{});
    int i = 0;
    for (; i + 4 <= nstages; i += 4) {
        issue(i + 3, hm_ic<3>{});
        stage(hm_ic<0>{});
        issue(i + 4, hm_ic<0>{});
        stage(hm_ic<1>{});
        issue(i + 5, hm_ic<1>{});
        stage(hm_ic<2>{});
        issue(i + 6, hm_ic<2>{});
        stage(hm_ic<3>{});
    }
    if (i < nstages) {
        issue(i + 3, hm_ic<3>{});
        stage(hm_ic<0>{});
        ++i;
    }
    if (i < nstages) {
        issue(i + 3, hm_ic<0>{});
        stage(hm_ic<1>{});
        ++i;
    }
    if (i < nstages) {
        issue(i + 3, hm_ic<1>{});
        stage(hm_ic<2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // drain the DMA queue before the ring becomes reduction scratch

    // ---- the four waves added in fixed order; register r of group grp of lane (q, n) is source q + 4 r, entry as in the
    //      float32 kernel.  A round = 4 groups; wave w adds the four values of register r = w of each.
    double* lds = reinterpret_cast<double*>(ring);
    const int NA = Mv * Mv;
    const int src = q + 4 * wave;
    double* vout = Vpart + (((size_t)blockIdx.y * F + f0) * K + src) * NA;
    const bool live = src < K;
#pragma unroll
    for (int g0 = 0; g0 < 17; g0 += 4) {
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (g0 + v < 17) lds[(v * 4 + r) * kHmLdsStride + tid] = acc[g0 + v][r];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int grp = g0 + v;                 // compile-time
            if (grp >= 17) continue;
            const int c = (grp + 1) >> 1, im = (grp + 1) & 1;            // grp 2c-1: re, 2c: im   (grp 0: the diagonal)
            double s = 0.;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += lds[(v * 4 + wave) * kHmLdsStride + w * 64 + lane];
            if (!live) continue;
            int pos;
            if (grp == 0) {
                if (n >= Mv) continue;
                pos = n;
            } else {
                if (c == MH && n >= MH) continue;                          // (c = M/2: the upper half of the lanes repeats the lower)
                const int mm = n + c >= M ? n + c - M : n + c;
                const int i = n < mm ? n : mm, j = n < mm ? mm : n;
                if (j >= Mv) continue;
                pos = herm_pair_index(Mv, i, j) + im;
                if (im && mm < n) s = -s;                                 // Im(x_i conj x_j) = -Im(x_j conj x_i)
            }
            vout[pos] = s;
        }
    }
}

}  // namespace

// 9..16 sources on 10/12/14/16 channels (odd counts: the padded copy of X).  Wt: the (T + 1, 16) table of final weights of
// launch_cov_weights, row T zeroed (kernels_cov_half16.hip fills it and calls this).
bool cov_hmfma_supported(int M, int K) { return M >= 10 && M <= 16 && M % 2 == 0 && K >= 9 && K <= 16; }

// float64 sums (`precise`): exactly 16 channels (15: the padded copy of X), 9..16 sources.  Wt: the (T + 1, 16) table of float64
// weights of kernels_cov_half16.hip (h64_weights_kernel), row T zero.
bool cov_hmfma64_supported(int M, int K) { return M == 16 && K >= 9 && K <= 16; }

hipError_t launch_cov_hmfma64(hipStream_t s, const float2* X, const double* Wt, double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g) {
    if (!cov_hmfma64_supported(M, K) || Mv > M || Mv < M - 1 || Wt == nullptr || g.tc % (8 * kHmFrames) != 0) return hipErrorInvalidValue;
    const char* flat = std::getenv("OIVA_HMFMA_FLAT");
    if (!(flat && flat[0] == '1') && (size_t)36 * F * M * 8 < 0xffffffffull)
        return launch_dominant(cov_hmfma64_kernel<true>, dim3(F, g.nsplit, 1), dim3(kBlock), 0, s, X, Wt, Vpart, T, F, Mv, K, g.tc);
    return launch_dominant(cov_hmfma64_kernel<false>, dim3(F, g.nsplit, 1), dim3(kBlock), 0, s, X, Wt, Vpart, T, F, Mv, K, g.tc);
}

hipError_t launch_cov_hmfma(hipStream_t s, const float2* X, const float* Wt, double* Vpart, int T, int F, int M, int Mv, int K, const CovGeom& g) {
    if (!cov_hmfma_supported(M, K) || Mv > M || Mv < M - 1 || Wt == nullptr || g.tc % (8 * kHmFrames) != 0) return hipErrorInvalidValue;
    const dim3 grid(F, g.nsplit, 1), block(kBlock);
    // the buffer form of the DMAs: a lane's offset from the first frame of a stage is 32 bits
    // ($OIVA_HMFMA_FLAT=1, read at every launch: the flat form whatever the size -- tests compare the two bit for bit)
    const char* flat = std::getenv("OIVA_HMFMA_FLAT");
    const bool buf = !(flat && flat[0] == '1') && (size_t)36 * F * M * 8 < 0xffffffffull;
    // ($OIVA_HMFMA_PK=0: the partners by DPP rotations of the lane's own operand, four vector instructions per product pair)
    const char* pkv = std::getenv("OIVA_HMFMA_PK");
    const bool pk = !(pkv && pkv[0] == '0');
    if (M == 16) {
        if (buf && pk) return launch_dominant(cov_hmfma_kernel<true, true, true>, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc, g.part32);
        if (buf) return launch_dominant(cov_hmfma_kernel<true, true, false>, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc, g.part32);
        if (pk) return launch_dominant(cov_hmfma_kernel<true, false, true>, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc, g.part32);
        return launch_dominant(cov_hmfma_kernel<true, false, false>, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc, g.part32);
    }
    if (buf) return launch_dominant(cov_hmfma_kernel<false, true, false>, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc, g.part32);
    return launch_dominant(cov_hmfma_kernel<false, false, false>, grid, block, 0, s, X, Wt, Vpart, T, F, M, Mv, K, g.tc, g.part32);
}

}  // namespace oiva
