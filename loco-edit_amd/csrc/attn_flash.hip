// Tangent and cotangent of multi-head self-attention WITHOUT per-probe [T x T] matrices (reference call sites:
// guided_diffusion/unet.py:330-356 QKVAttentionLegacy under jacfwd / autograd.functional.jacobian, edit.py:2447-2480).
//
// The generic path (engine.hip + gemm.hip + softmax_jac) writes a T x T tangent score matrix per (probe, head) and passes
// over it four times; at 1024 tokens that traffic, not arithmetic, is what the attention blocks cost.  Here a workgroup
// owns 128 tokens of one (probe, head) and streams over the other token axis in blocks of 64, keeping the score tile in
// registers: only the primal probabilities P (B = 1, shared by every probe, cache resident) are read as a matrix.
//
//   tangent      do = (dP) v^T + P dv^T,   dP = s P o (dS - rowsum(P o dS)),   dS = dq^T k + q^T dk
//                =>  do_i = sum_j [ W_ij v_j + P_ij dv_j ] - r_i o_i,   W = s P o dS,   r_i = sum_j W_ij      (o = P v, primal)
//   cotangent    g_S = s P o (g_P - delta_i),  g_P = g_o^T v,  delta_i = <g_o_i, o_i>
//                g_q_i = sum_j g_S_ij k_j   (rows owned: COTQ)      g_k_j = sum_i g_S_ij q_i,  g_v_j = sum_i P_ij g_o_i   (columns owned: COTK)
//
// Arithmetic: split-bf16 (hi + lo halves, three v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate) like the
// convolutions.  Both GEMMs of a block are oriented so that the first one's D fragment (rows = streamed tokens, column =
// the lane's own token) IS the second one's B operand: a 32x32 D fragment holds rows {0-3, 8-11, 16-19, 24-27} + 4*khalf
// in its 16 registers, i.e. registers 8b..8b+7 are the 16 streamed tokens of k-step b in the order
// j = (t&3) + 8(t>>2) + 4*khalf -- a permutation of the contraction index, which the A operand of the second GEMM
// (records written by THIS kernel's staging) simply adopts.  No LDS round trip for the score tile.
//
// Heads of up to 64 channels (the ADM / IF-shaped denoisers' 64, Stable Diffusion's 40 at its 4096-token level: narrower
// heads run zero-padded to 64; two workgroups per CU) and, round 4, up to 96 channels (Stable Diffusion v1's 80-channel heads
// at its 1024-token level: three 32-channel output tiles, one workgroup per CU); token counts that are multiples of 128.
// Other shapes stay on the generic path.
//
// TXT (the DeepFloyd-IF attention, engine.hip "attention over [text ; image] keys"): the key axis is [Lt text states ; T
// image tokens] under ONE softmax, P is [T][Lt + T] with the text columns first.  The text keys / values (a.kt / a.vt,
// [CH][Lt] per head, zero beyond the real states) are constants of the prompt: they are streamed as Lt / 64 extra key blocks
// ahead of the image blocks with zero key / value tangents (TAN) and take part in g_q (COTQ); g_k / g_v exist for the
// image keys only, so COTK just reads its column of P at Lt + j.
#include "kernels.h"
#include <cstdlib>
#include <cstring>

namespace loco {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
typedef short s16x8a __attribute__((ext_vector_type(8)));

namespace {

constexpr int RP = 80;            // record pitch: 64-byte record [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] + 16 (conflict-free b128 reads)
constexpr int CHD = 96;           // widest head (channels): NCT = 2 channel tiles of 32 up to 64 channels, 3 up to 96
constexpr int NOWN = 128;         // own tokens per workgroup (32 per wave)
constexpr int NBLK = 64;          // streamed tokens per block
// bytes of one operand region for heads padded to 32 * NCT channels: 2 tensors x 2 NCT records (16 channels each) x 64 rows
constexpr int region_bytes(int nct) { return 2 * (2 * nct) * NBLK * RP; }

// two fp32 values -> one dword of bf16 hi parts and one of the residuals' bf16 values: ONE v_cvt_pk_bf16_f32 each (the scalar
// __bf16 casts compile to a conversion per value plus shift / or packing: 11 vector instructions per pair instead of 6; same
// roundings, same bits -- conv_bf16_kernel.h cvt2)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xffff0000u);
    float da = a - ha;
    asm volatile("" : "+v"(da));
    const float db = b - hb;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){da, db}, bf16x2_));
}
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
    split2(v[0], v[1], hi.x, lo.x);
    split2(v[2], v[3], hi.y, lo.y);
    split2(v[4], v[5], hi.z, lo.z);
    split2(v[6], v[7], hi.w, lo.w);
}

// token-major records of X[c][t] (row stride T): tokens t0 .. t0+ntok-1, 64 channels -> dst[(chunk16 * ntok + tok) * RP].
// Load and store halves are separate so the streamed block t+1 is in flight (registers) while block t multiplies.
// NOCT = octets (8 channels) of the padded head = 4 * NCT
template <int NTOK, int NOCT> struct TokRegs { float v[(NTOK * NOCT) / 256][8]; };
template <int NTOK, int NOCT>
__device__ __forceinline__ void load_tokens(TokRegs<NTOK, NOCT>& R, const float* X, int T, int t0, int tid, int nch) {
#pragma unroll
    for (int it = 0; it < (NTOK * NOCT) / 256; ++it) {
        const int e = tid + it * 256;
        const int tok = e % NTOK, oct = e / NTOK;              // lanes run over tokens: coalesced 4-byte loads
#pragma unroll
        for (int k = 0; k < 8; ++k)                            // heads narrower than 64 channels: the rest of the K dimension is zero
            R.v[it][k] = (oct * 8 + k < nch) ? X[(long)(oct * 8 + k) * T + t0 + tok] : 0.f;
    }
}
template <int NTOK, int NOCT>
__device__ __forceinline__ void zero_tokens(TokRegs<NTOK, NOCT>& R) {
#pragma unroll
    for (int it = 0; it < (NTOK * NOCT) / 256; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k) R.v[it][k] = 0.f;
}
template <int NTOK, int NOCT>
__device__ __forceinline__ void store_tokens(const TokRegs<NTOK, NOCT>& R, unsigned char* dst, int tid) {
#pragma unroll
    for (int it = 0; it < (NTOK * NOCT) / 256; ++it) {
        const int e = tid + it * 256;
        const int tok = e % NTOK, oct = e / NTOK;
        uint4 hi, lo;
        split8(R.v[it], hi, lo);
        unsigned char* r = dst + ((oct >> 1) * NTOK + tok) * RP + (oct & 1) * 16;
        *reinterpret_cast<uint4*>(r) = hi;
        *reinterpret_cast<uint4*>(r + 32) = lo;
    }
}
template <int NTOK, int NOCT>
__device__ __forceinline__ void stage_tokens(const float* X, int T, int t0, unsigned char* dst, int tid, int nch) {
    TokRegs<NTOK, NOCT> R;
    load_tokens<NTOK, NOCT>(R, X, T, t0, tid, nch);
    store_tokens<NTOK, NOCT>(R, dst, tid);
}
// channel-major records of X[c][t]: CHP = 32 NCT channels x tokens t0 .. t0+63 in 4 blocks of 16, the 16 tokens of a record
// in the D-fragment order j = (t&3) + 8(t>>2) + 4*khalf (slot = khalf*8 + t) -> dst[(jb * CHP + c) * RP]; the 256 threads
// cover 64 channels per pass (4 lanes = 64 consecutive floats of one row), NCP = ceil(CHP / 64) passes
template <int NCP> struct ChRegs { f32x4a a[NCP][4]; };
template <int NCP>
__device__ __forceinline__ void load_channels(ChRegs<NCP>& R, const float* X, int T, int t0, int tid, int nch) {
    const int jb = tid & 3;
#pragma unroll
    for (int ps = 0; ps < NCP; ++ps) {
        const int c = ps * 64 + (tid >> 2);
        if (c < nch) {
            const f32x4a* p = reinterpret_cast<const f32x4a*>(X + (long)c * T + t0 + jb * 16);
            R.a[ps][0] = p[0]; R.a[ps][1] = p[1]; R.a[ps][2] = p[2]; R.a[ps][3] = p[3];   // tokens 0-3, 4-7, 8-11, 12-15
        } else {
            R.a[ps][0] = R.a[ps][1] = R.a[ps][2] = R.a[ps][3] = f32x4a{0.f, 0.f, 0.f, 0.f};
        }
    }
}
template <int NCP>
__device__ __forceinline__ void zero_channels(ChRegs<NCP>& R) {
#pragma unroll
    for (int ps = 0; ps < NCP; ++ps) R.a[ps][0] = R.a[ps][1] = R.a[ps][2] = R.a[ps][3] = f32x4a{0.f, 0.f, 0.f, 0.f};
}
template <int NCP, int CHP>
__device__ __forceinline__ void store_channels(const ChRegs<NCP>& R, unsigned char* dst, int tid) {
    const int jb = tid & 3;
#pragma unroll
    for (int ps = 0; ps < NCP; ++ps) {
        const int c = ps * 64 + (tid >> 2);
        if (c >= CHP) continue;                                // the last pass of a 96-channel head: half of the threads
        const f32x4a* a = R.a[ps];
        const float k0[8] = {a[0][0], a[0][1], a[0][2], a[0][3], a[2][0], a[2][1], a[2][2], a[2][3]};   // khalf 0: 0-3, 8-11
        const float k1[8] = {a[1][0], a[1][1], a[1][2], a[1][3], a[3][0], a[3][1], a[3][2], a[3][3]};   // khalf 1: 4-7, 12-15
        uint4 h0, l0, h1, l1;
        split8(k0, h0, l0);
        split8(k1, h1, l1);
        unsigned char* r = dst + (jb * CHP + c) * RP;
        *reinterpret_cast<uint4*>(r) = h0;
        *reinterpret_cast<uint4*>(r + 16) = h1;
        *reinterpret_cast<uint4*>(r + 32) = l0;
        *reinterpret_cast<uint4*>(r + 48) = l1;
    }
}

struct Frag { s16x8a hi, lo; };
__device__ __forceinline__ Frag ld_frag(const unsigned char* rec, int khalf) {
    Frag f;
    f.hi = *reinterpret_cast<const s16x8a*>(rec + khalf * 16);
    f.lo = *reinterpret_cast<const s16x8a*>(rec + 32 + khalf * 16);
    return f;
}
__device__ __forceinline__ void mma3(f32x16& acc, const Frag& a, const Frag& b) {
    const bf16x8a ah = __builtin_bit_cast(bf16x8a, a.hi), al = __builtin_bit_cast(bf16x8a, a.lo);
    const bf16x8a bh = __builtin_bit_cast(bf16x8a, b.hi), bl = __builtin_bit_cast(bf16x8a, b.lo);
#if defined(AF_WI) && (AF_WI & 16)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
#endif
}
// registers 8b .. 8b+7 of a D fragment as the B operand of k-step b
__device__ __forceinline__ Frag frag_of(const f32x16& d, int b) {
    float v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = d[8 * b + t];
    uint4 hi, lo;
    split8(v, hi, lo);
    Frag f;
    f.hi = __builtin_bit_cast(s16x8a, hi);
    f.lo = __builtin_bit_cast(s16x8a, lo);
    return f;
}

enum : int { M_TAN = 0, M_COTQ = 1, M_COTK = 2 };

// NCK: 16-channel k-steps of the score products that hold channels (heads of <= 48 channels skip the all-zero fourth
// step at compile time; a run-time skip splits the block's scheduling region and measured 3.4 % slower)
// NCT: 32-channel output tiles of the padded head (2: heads up to 64 channels, two workgroups per CU; 3: up to 96 channels
// -- Stable Diffusion v1's 80-channel heads at its 1024-token level -- one workgroup per CU: 120 KB of operand regions)
#ifndef AF_WI
#define AF_WI 0
#endif
template <int MODE, int NCK, int NCT, bool TXT = false>
__global__ __launch_bounds__(256, (AF_WI & 8) ? 2 : 1) void attn_flash_kernel(AttnFlashArgs a) {
    constexpr int NREC = 2 * NCT;                  // 16-channel records per token
    constexpr int NOCT = 4 * NCT;                  // octets per token
    constexpr int CHP = 32 * NCT;                  // padded head width
    constexpr int NCP = (CHP + 63) / 64;
    constexpr int REGION = region_bytes(NCT);
    constexpr int T2 = NREC * NBLK * RP;           // offset of the second tensor inside a region
    static_assert(NCK <= NREC, "k-steps of the score product");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const RA = lds;                 // token-major operands of the streamed block (GEMM 1)
    unsigned char* const RB = lds + REGION;        // channel-major operands of the streamed block (GEMM 2)
    float* const DL = reinterpret_cast<float*>(lds + T2); // COTK: delta of the 64 streamed queries (RA holds one tensor there)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, khalf = lane >> 5;
    const int T = a.T, h = blockIdx.y, b = blockIdx.z;
    const int own0 = blockIdx.x * NOWN;            // first own token of the workgroup
    const int mytok = own0 + wave * 32 + l31;      // the lane's own token (column of every D fragment)
    const int nch = a.CH;                          // head width (<= CHP; narrower heads run zero-padded)
    const long HS = a.hs, HO = (long)nch * T;
    const float* q = a.q + h * HS;  const float* k = a.k + h * HS;  const float* v = a.v + h * HS;
    const int Lt = TXT ? a.Lt : 0;                 // text columns ahead of the image columns in every row of P
    const int PS = Lt + T;                         // row stride of P
    const float* P = a.P + (long)h * T * PS;
    const float* kt = TXT ? a.kt + (long)h * nch * Lt : nullptr;
    const float* vt = TXT ? a.vt + (long)h * nch * Lt : nullptr;
    const float* o = a.o + h * HO;
    const float* dq = MODE == M_TAN ? a.dq + (long)b * a.bs_d + h * HS : nullptr;
    const float* dk = MODE == M_TAN ? a.dk + (long)b * a.bs_d + h * HS : nullptr;
    const float* dv = MODE == M_TAN ? a.dv + (long)b * a.bs_d + h * HS : nullptr;
    const float* go = MODE != M_TAN ? a.go + (long)b * a.bs_go + h * HO : nullptr;

    // ---- own-token B fragments (constant over the stream): staged once through the operand regions
    Frag y1[NREC], y2[MODE == M_TAN ? NREC : 1];
    {
        const float* Y1 = MODE == M_TAN ? dq : (MODE == M_COTQ ? go : v);
        stage_tokens<NOWN, NOCT>(Y1, T, own0, lds, tid, nch);
        if (MODE == M_TAN) stage_tokens<NOWN, NOCT>(q, T, own0, lds + NREC * NOWN * RP, tid, nch);
        __syncthreads();
#pragma unroll
        for (int ck = 0; ck < NREC; ++ck) {
            y1[ck] = ld_frag(lds + (ck * NOWN + wave * 32 + l31) * RP, khalf);
            if (MODE == M_TAN) y2[ck] = ld_frag(lds + NREC * NOWN * RP + (ck * NOWN + wave * 32 + l31) * RP, khalf);
        }
    }
    float delta_own = 0.f;                          // COTQ: delta_i = <g_o_i, o_i> of the lane's own query
    if (MODE == M_COTQ) {
        float s = 0.f;
        for (int c = khalf * (CHP / 2); c < khalf * (CHP / 2) + CHP / 2 && c < nch; ++c) s += go[(long)c * T + mytok] * o[(long)c * T + mytok];
        s += __shfl_xor(s, 32, 64);
        delta_own = s;
        if (khalf == 0) a.delta[((long)b * a.NH + h) * T + mytok] = s;
    }

    f32x16 acc[NCT], acc2[MODE == M_COTK ? NCT : 1];    // [c tile]: GEMM 2 accumulators (COTK: g_v and g_k)
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; if (MODE == M_COTK) acc2[i][r] = 0.f; }
    float rsum = 0.f;                               // TAN: r_i partial of this lane (its khalf's rows)

    // streamed operands of a block: loaded into registers one block ahead, converted into LDS records at the top of the block
    TokRegs<NBLK, NOCT> ta, tb;
    ChRegs<NCP> ca, cb;
    float dreg = 0.f;
    // u: position on the streamed axis.  TAN / COTQ stream the keys, [text ; image] under TXT (u < Lt: a text block);
    // COTK streams the queries (image tokens only)
    const int U = (TXT && MODE != M_COTK) ? PS : T;
    auto fetch = [&](int u) {
        const bool text = TXT && MODE != M_COTK && u < Lt;
        const int t0 = (TXT && MODE != M_COTK) ? u - Lt : u;
        if (MODE == M_TAN) {
            if (text) {
                load_tokens<NBLK, NOCT>(ta, kt, Lt, u, tid, nch); zero_tokens<NBLK, NOCT>(tb);
                load_channels<NCP>(ca, vt, Lt, u, tid, nch); zero_channels<NCP>(cb);
                return;
            }
            load_tokens<NBLK, NOCT>(ta, k, T, t0, tid, nch); load_tokens<NBLK, NOCT>(tb, dk, T, t0, tid, nch);
            load_channels<NCP>(ca, v, T, t0, tid, nch); load_channels<NCP>(cb, dv, T, t0, tid, nch);
        } else if (MODE == M_COTQ) {
            if (text) {
                load_tokens<NBLK, NOCT>(ta, vt, Lt, u, tid, nch);
                load_channels<NCP>(ca, kt, Lt, u, tid, nch);
                return;
            }
            load_tokens<NBLK, NOCT>(ta, v, T, t0, tid, nch);
            load_channels<NCP>(ca, k, T, t0, tid, nch);
        } else {
            load_tokens<NBLK, NOCT>(ta, go, T, t0, tid, nch);
            load_channels<NCP>(ca, go, T, t0, tid, nch); load_channels<NCP>(cb, q, T, t0, tid, nch);
            if (tid < NBLK) dreg = a.delta[((long)b * a.NH + h) * T + t0 + tid];
        }
    };
    fetch(0);
    for (int t0 = 0; t0 < U; t0 += NBLK) {
        __syncthreads();                            // the previous block's fragment reads (and the prologue's) are done
        if (!(AF_WI & 4) || t0 == 0) {
        store_tokens<NBLK, NOCT>(ta, RA, tid);
        store_channels<NCP, CHP>(ca, RB, tid);
        if (MODE == M_TAN) { store_tokens<NBLK, NOCT>(tb, RA + T2, tid); store_channels<NCP, CHP>(cb, RB + T2, tid); }
        if (MODE == M_COTK) { store_channels<NCP, CHP>(cb, RB + T2, tid); if (tid < NBLK) DL[tid] = dreg; }
        }
        if (!(AF_WI & 2) && t0 + NBLK < U) fetch(t0 + NBLK);        // in flight under this block's MFMAs
        // primal probabilities of the tile in the D-fragment layout (rows = streamed tokens, column = own token)
        f32x16 pt[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (AF_WI & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) pt[mt][r] = a.scale;
            } else
            if (MODE != M_COTK) {                   // own = query i (a row of P): 4 consecutive streamed keys per load
                const float* pr = P + (long)mytok * PS + t0 + 32 * mt + 4 * khalf;      // t0 is the column of P here (text first)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4a x = *reinterpret_cast<const f32x4a*>(pr + 8 * qd);
                    pt[mt][4 * qd] = x[0]; pt[mt][4 * qd + 1] = x[1]; pt[mt][4 * qd + 2] = x[2]; pt[mt][4 * qd + 3] = x[3];
                }
            } else {                                // own = key j (a column of P): streamed queries are the rows
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = t0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    pt[mt][r] = P[(long)i * PS + Lt + mytok];
                }
            }
        }
        __syncthreads();
        // ---- GEMM 1: S[streamed token][own token], K = 64 channels (TAN: two products)
        f32x16 s_[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_[mt][r] = 0.f;
#pragma unroll
            for (int ck = 0; ck < NCK; ++ck) {
                mma3(s_[mt], ld_frag(RA + (ck * NBLK + 32 * mt + l31) * RP, khalf), y1[ck]);
                if (MODE == M_TAN) mma3(s_[mt], ld_frag(RA + T2 + (ck * NBLK + 32 * mt + l31) * RP, khalf), y2[ck]);
            }
        }
        // ---- elementwise: W = s P o S (TAN), G = s P o (S - delta) (COT)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float w;
                if (MODE == M_TAN) { w = a.scale * pt[mt][r] * s_[mt][r]; rsum += w; }
                else if (MODE == M_COTQ) w = a.scale * pt[mt][r] * (s_[mt][r] - delta_own);
                else w = a.scale * pt[mt][r] * (s_[mt][r] - DL[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * khalf]);
                s_[mt][r] = w;
            }
        // ---- GEMM 2: acc[channel][own token] += sum over the 64 streamed tokens (4 k-steps of 16, fragment order)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int jb = mt * 2 + bb;
                const Frag fw = (AF_WI & 4) ? y1[0] : frag_of(s_[mt], bb);
                Frag fp;
                if (MODE != M_COTQ) fp = (AF_WI & 4) ? y1[1] : frag_of(pt[mt], bb);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const unsigned char* r1 = RB + (jb * CHP + 32 * ct + l31) * RP;
                    if (MODE == M_TAN) {
                        mma3(acc[ct], ld_frag(r1, khalf), fw);                           // v W
                        mma3(acc[ct], ld_frag(r1 + T2, khalf), fp);                      // dv P
                    } else if (MODE == M_COTQ) {
                        mma3(acc[ct], ld_frag(r1, khalf), fw);                           // k G
                    } else {
                        mma3(acc[ct], ld_frag(r1, khalf), fp);                           // g_o P  -> g_v
                        mma3(acc2[ct], ld_frag(r1 + T2, khalf), fw);                     // q G    -> g_k
                    }
                }
            }
    }
    // ---- epilogue.  D[row = channel][col = own token]
    if (MODE == M_TAN) rsum += __shfl_xor(rsum, 32, 64);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (c >= nch) continue;
            const long off = (long)c * T + mytok;
            if (MODE == M_TAN) {
                a.out[(long)b * a.bs_out + h * HO + off] = acc[ct][r] - rsum * o[off];
            } else if (MODE == M_COTQ) {
                a.gq[(long)b * a.bs_g + h * HS + off] = acc[ct][r];
            } else {
                a.gv[(long)b * a.bs_g + h * HS + off] = acc[ct][r];
                a.gk[(long)b * a.bs_g + h * HS + off] = acc2[ct][r];
            }
        }
}


// =====================================================================================================================
// Round 5: the same three kernels with every MFMA operand delivered by LDS-DMA from PRE-SPLIT records.
//
// What the what-if timing of the kernel above showed (tests/diag/attn_bench.hip, 4096 tokens x 8 heads x 40 channels, 5 probes,
// tangent 1 922 us): without the streamed operands' global loads 1 229 us, without their conversion + ds_write 1 658 us, without the
// loads of P 1 433 us, without all three 705 us (of which ~570 us are the MFMAs): two thirds of the launch were exposed memory
// latency and staging -- the tangent and key-side cotangent kernels hold 323 / 338 registers, i.e. ONE wave per SIMD, and every
// workgroup of a (probe, head) re-converted the whole k / v / dk / dv (32 query tiles: 32 x the conversion work).
//   * `attn_split_kernel` converts each operand tensor ONCE per launch into the two record layouts the MFMAs read (token-major
//     for the score products, channel-major in D-fragment token order for the output products; 64-byte records, 16-byte pieces
//     XOR-swizzled by (record >> 2) & 3 so that the ds_read_b128 of 16 consecutive records hit 16 disjoint bank groups);
//   * the attention kernel streams those records global -> LDS by `global_load_lds_dwordx4` (no staging registers, no conversion,
//     no ds_write): the token-major region is refilled for block t+1 as soon as block t's score products have read it (under the
//     output products of t), the channel-major region as soon as block t's output products are done (under the score products of
//     t+1); one raw s_barrier per phase, counted vmcnt waits;
//   * the own-token fragments come straight from the records; the P tile of block t+1 is loaded into the registers block t frees
//     while its output products run;
//   * 64 staging registers less: <= 256 registers, TWO workgroups per CU for heads up to 64 channels;
//   * the probes of one (query tile, head) are adjacent in dispatch order on one XCD (blocks L, L + 8, ...): the tile's rows of P
//     are fetched from HBM once per launch, not once per probe (537 MB per head set at 4096 tokens).
// Same products in the same order as the kernel above: bit-identical results (tests/diag/attn_bench.hip compares the two).
// The kernel above stays as the fallback when the caller gives no workspace (AttnFlashArgs::ws) or LOCO_FLASH_DMA=0.

constexpr int RSZ = 64;                                       // record bytes
__device__ __forceinline__ int sw_off(int rec, int piece) { return rec * RSZ + ((piece ^ ((rec >> 2) & 3)) << 4); }

struct SplitJob { const float* src; long bs, hs; int nb, T, layout; unsigned char* dst; };       // layout 0: token-major, 1: channel-major
struct SplitJobs { SplitJob j[8]; int n, NCK, CHP, nch, NH; };

// grid (max T / 64, NH, sum of nb): one 64-token block of one (sample, head) of one job
__global__ __launch_bounds__(256) void attn_split_kernel(SplitJobs J) {
    int z = blockIdx.z, ji = 0;
    while (ji < J.n - 1 && z >= J.j[ji].nb) { z -= J.j[ji].nb; ++ji; }
    const SplitJob jb_ = J.j[ji];
    const int nblk = jb_.T / NBLK, blk = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, Tr = jb_.T, nch = J.nch;
    if (blk >= nblk) return;
    const float* X = jb_.src + (long)z * jb_.bs + (long)h * jb_.hs;
    const int t0 = blk * NBLK;
    if (jb_.layout == 0) {
        unsigned char* dst = jb_.dst + (((long)z * J.NH + h) * nblk + blk) * ((long)J.NCK * 4096);
        for (int e = tid; e < NBLK * 2 * J.NCK; e += 256) {
            const int tok = e % NBLK, oct = e / NBLK;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (oct * 8 + k < nch) ? X[(long)(oct * 8 + k) * Tr + t0 + tok] : 0.f;
            uint4 hi, lo;
            split8(v, hi, lo);
            const int rec = (oct >> 1) * NBLK + tok;
            *reinterpret_cast<uint4*>(dst + sw_off(rec, oct & 1)) = hi;
            *reinterpret_cast<uint4*>(dst + sw_off(rec, 2 + (oct & 1))) = lo;
        }
    } else {
        unsigned char* dst = jb_.dst + (((long)z * J.NH + h) * nblk + blk) * ((long)J.CHP * 256);
        const int jb = tid & 3;
        for (int c = tid >> 2; c < J.CHP; c += 64) {
            f32x4a a[4];
            if (c < nch) {
                const f32x4a* p = reinterpret_cast<const f32x4a*>(X + (long)c * Tr + t0 + jb * 16);
                a[0] = p[0]; a[1] = p[1]; a[2] = p[2]; a[3] = p[3];
            } else {
                a[0] = a[1] = a[2] = a[3] = f32x4a{0.f, 0.f, 0.f, 0.f};
            }
            const float k0[8] = {a[0][0], a[0][1], a[0][2], a[0][3], a[2][0], a[2][1], a[2][2], a[2][3]};   // khalf 0: tokens 0-3, 8-11
            const float k1[8] = {a[1][0], a[1][1], a[1][2], a[1][3], a[3][0], a[3][1], a[3][2], a[3][3]};   // khalf 1: 4-7, 12-15
            uint4 h0, l0, h1, l1;
            split8(k0, h0, l0);
            split8(k1, h1, l1);
            const int rec = jb * J.CHP + c;
            *reinterpret_cast<uint4*>(dst + sw_off(rec, 0)) = h0;
            *reinterpret_cast<uint4*>(dst + sw_off(rec, 1)) = h1;
            *reinterpret_cast<uint4*>(dst + sw_off(rec, 2)) = l0;
            *reinterpret_cast<uint4*>(dst + sw_off(rec, 3)) = l1;
        }
    }
}

struct FlashRecs {
    const unsigned char *own1, *own2; long own1_bs, own2_bs;   // token-major records of the own-token operands (sample stride; 0: primal)
    const unsigned char *sa1, *sa2; long sa1_bs, sa2_bs;       // streamed, token-major (A of the score products)
    const unsigned char *sb1, *sb2; long sb1_bs, sb2_bs;       // streamed, channel-major (A of the output products)
    const unsigned char *ta1, *tb1;                            // TXT: the prompt's text blocks in the two layouts (constants)
    const unsigned char* zero;                                 // a block of zero records (tangents of those constants)
};

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef const __attribute__((address_space(1))) unsigned char glb_u8;

template <int MODE, int NCK, int NCT, bool TXT = false>
__global__ __launch_bounds__(256, NCT == 2 ? 2 : 1) void attn_flash_dma_kernel(AttnFlashArgs a, FlashRecs R) {
    constexpr int CHP = 32 * NCT;
    constexpr int ABLK = NCK * 4096;               // bytes of a token-major block (64 tokens x NCK records)
    constexpr int BBLK = CHP * 256;                // bytes of a channel-major block (4 k-steps x CHP rows)
    constexpr int NTA = MODE == M_TAN ? 2 : 1;     // streamed token-major tensors
    constexpr int NTB = MODE == M_COTQ ? 1 : 2;    // streamed channel-major tensors
    constexpr int NPL = MODE == M_COTK ? 33 : 8;   // plain loads per thread and block (P tile; COTK: + delta)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const RA = lds;
    unsigned char* const RB = lds + NTA * ABLK;
    float* const DL = reinterpret_cast<float*>(lds + NTA * ABLK + NTB * BBLK);      // COTK: delta of the streamed queries, one copy per wave
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, nch = a.CH;
    int tile, b;
    {
        const int NTT = (T / NOWN) * a.NH, L = blockIdx.x;
        if ((NTT & 7) == 0) { const int q_ = L >> 3; b = q_ % a.B; tile = (q_ / a.B) * 8 + (L & 7); }
        else { b = L % a.B; tile = L / a.B; }
        tile = __builtin_amdgcn_readfirstlane(tile);
        b = __builtin_amdgcn_readfirstlane(b);
    }
    const int h = tile / (T / NOWN), own0 = (tile % (T / NOWN)) * NOWN;
    const int mytok = own0 + wave * 32 + l31;
    const long HS = a.hs, HO = (long)nch * T;
    const int Lt = TXT ? a.Lt : 0;
    const int PS = Lt + T;
    const float* P = a.P + (long)h * T * PS;
    const float* o = a.o + h * HO;
    const float* go = MODE != M_TAN ? a.go + (long)b * a.bs_go + h * HO : nullptr;
    const int nblk = T / NBLK, ntxt = Lt / NBLK;
    // record bases of this (probe, head)
    const long hsA = (long)nblk * ABLK, hsB = (long)nblk * BBLK;
    const unsigned char* const own1 = R.own1 + (long)b * R.own1_bs + h * hsA;
    const unsigned char* const own2 = MODE == M_TAN ? R.own2 + (long)b * R.own2_bs + h * hsA : nullptr;
    const unsigned char* const sa1 = R.sa1 + (long)b * R.sa1_bs + h * hsA;
    const unsigned char* const sa2 = NTA == 2 ? R.sa2 + (long)b * R.sa2_bs + h * hsA : nullptr;
    const unsigned char* const sb1 = R.sb1 + (long)b * R.sb1_bs + h * hsB;
    const unsigned char* const sb2 = NTB == 2 ? R.sb2 + (long)b * R.sb2_bs + h * hsB : nullptr;
    const unsigned char* const ta1 = TXT ? R.ta1 + (long)h * ntxt * ABLK : nullptr;
    const unsigned char* const tb1 = TXT ? R.tb1 + (long)h * ntxt * BBLK : nullptr;

    // the lane's piece offsets inside a run of 32 records (row l31): hi / lo piece of its k-half, swizzled
    const int kx = (l31 >> 2) & 3;
    const unsigned offh = (unsigned)l31 * RSZ + (unsigned)((khalf ^ kx) << 4), offl = (unsigned)l31 * RSZ + (unsigned)(((2 + khalf) ^ kx) << 4);

    // ---- own-token B fragments straight from the records
    Frag y1[NCK], y2[MODE == M_TAN ? NCK : 1];
    {
        const long ob = (long)((own0 >> 6) + (wave >> 1)) * ABLK + (wave & 1) * (32 * RSZ);
#pragma unroll
        for (int ck = 0; ck < NCK; ++ck) {
            y1[ck].hi = *reinterpret_cast<const s16x8a*>(own1 + ob + ck * 4096 + offh);
            y1[ck].lo = *reinterpret_cast<const s16x8a*>(own1 + ob + ck * 4096 + offl);
            if (MODE == M_TAN) {
                y2[ck].hi = *reinterpret_cast<const s16x8a*>(own2 + ob + ck * 4096 + offh);
                y2[ck].lo = *reinterpret_cast<const s16x8a*>(own2 + ob + ck * 4096 + offl);
            }
        }
    }
    float delta_own = 0.f;
    if (MODE == M_COTQ) {
        float s = 0.f;
        for (int c = khalf * (CHP / 2); c < khalf * (CHP / 2) + CHP / 2 && c < nch; ++c) s += go[(long)c * T + mytok] * o[(long)c * T + mytok];
        s += __shfl_xor(s, 32, 64);
        delta_own = s;
        if (khalf == 0) a.delta[((long)b * a.NH + h) * T + mytok] = s;
    }
    // A "use" of the own fragments in front of the loop: the compiler then waits for their loads HERE.  Left pending into the loop
    // they cost an s_waitcnt vmcnt(0) in front of the first MFMA of every block (the waitcnt pass cannot count across the back
    // edge), which also waited for the DMAs that had just been issued: no overlap of RB(n)'s transfer with the score products.
#pragma unroll
    for (int ck = 0; ck < NCK; ++ck) {
        asm volatile("" :: "v"(y1[ck].hi), "v"(y1[ck].lo));
        if (MODE == M_TAN) asm volatile("" :: "v"(y2[ck].hi), "v"(y2[ck].lo));
    }

    f32x16 acc[NCT], acc2[MODE == M_COTK ? NCT : 1];
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; if (MODE == M_COTK) acc2[i][r] = 0.f; }
    float rsum = 0.f;

    const int NB_ = (TXT && MODE != M_COTK) ? ntxt + nblk : nblk;       // streamed blocks ([text ; image] under TXT)
    // DMA of streamed block n: the token-major region / the channel-major region
    auto dma_A = [&](int n) {
        const bool text = TXT && MODE != M_COTK && n < ntxt;
        const int bi = __builtin_amdgcn_readfirstlane(text ? n : n - ((TXT && MODE != M_COTK) ? ntxt : 0));
        const unsigned char* s1 = text ? ta1 + (long)bi * ABLK : sa1 + (long)bi * ABLK;
#pragma unroll
        for (int i = 0; i < NCK; ++i)
            __builtin_amdgcn_global_load_lds((glb_u8*)(s1 + i * 4096 + tid * 16), (lds_u8*)(RA + i * 4096 + wave * 1024), 16, 0, 0);
        if constexpr (NTA == 2) {
            const unsigned char* s2 = text ? R.zero : sa2 + (long)bi * ABLK;
#pragma unroll
            for (int i = 0; i < NCK; ++i)
                __builtin_amdgcn_global_load_lds((glb_u8*)(s2 + i * 4096 + tid * 16), (lds_u8*)(RA + ABLK + i * 4096 + wave * 1024), 16, 0, 0);
        }
    };
    auto dma_B = [&](int n) {
        const bool text = TXT && MODE != M_COTK && n < ntxt;
        const int bi = __builtin_amdgcn_readfirstlane(text ? n : n - ((TXT && MODE != M_COTK) ? ntxt : 0));
        const unsigned char* s1 = text ? tb1 + (long)bi * BBLK : sb1 + (long)bi * BBLK;
#pragma unroll
        for (int i = 0; i < CHP / 16; ++i)
            __builtin_amdgcn_global_load_lds((glb_u8*)(s1 + i * 4096 + tid * 16), (lds_u8*)(RB + i * 4096 + wave * 1024), 16, 0, 0);
        if constexpr (NTB == 2) {
            const unsigned char* s2 = text ? R.zero : sb2 + (long)bi * BBLK;
#pragma unroll
            for (int i = 0; i < CHP / 16; ++i)
                __builtin_amdgcn_global_load_lds((glb_u8*)(s2 + i * 4096 + tid * 16), (lds_u8*)(RB + BBLK + i * 4096 + wave * 1024), 16, 0, 0);
        }
    };
    // primal probabilities of block n, 32-token half mt, in the D-fragment layout (rows = streamed tokens, column = own token)
    f32x16 pt[2];
    const unsigned p_lane = MODE != M_COTK ? ((unsigned)mytok * (unsigned)PS + 4u * khalf) * 4u
                                           : ((unsigned)(4 * khalf) * (unsigned)PS + (unsigned)(Lt + mytok)) * 4u;
    auto load_P = [&](int n, int mt) {
        const int u = n * NBLK;
        if (AF_WI & 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[mt][r] = a.scale;
            return;
        }
        // wave-uniform base (the block's first column / row of P) + a 32-bit lane offset: scalar-base addressing, no 64-bit
        // address arithmetic per load
        if (MODE != M_COTK) {
            const unsigned char* pb = reinterpret_cast<const unsigned char*>(P + u);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4a x = *reinterpret_cast<const f32x4a*>(pb + (p_lane + (unsigned)(128 * mt + 32 * qd)));
                pt[mt][4 * qd] = x[0]; pt[mt][4 * qd + 1] = x[1]; pt[mt][4 * qd + 2] = x[2]; pt[mt][4 * qd + 3] = x[3];
            }
        } else {
            // delta of the block's 64 queries: by DMA into the wave's OWN 256 bytes of LDS (no register, no cross-wave hand-over:
            // the elementwise step of block n-1 is behind this wave, the wait in front of block n's covers the transfer)
            if (mt == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) float*)(a.delta + ((long)b * a.NH + h) * T + u + lane),
                                                 (__attribute__((address_space(3))) float*)(DL + wave * 64), 4, 0, 0);
            const unsigned char* pb = reinterpret_cast<const unsigned char*>(P + (long)u * PS);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned ro = (unsigned)((32 * mt + (r & 3) + 8 * (r >> 2)) * PS) * 4u;       // (wave-uniform)
                pt[mt][r] = *reinterpret_cast<const float*>(pb + (p_lane + ro));
            }
        }
    };

    // ---- prologue: block 0's token-major records, then its P tile (this order is what the counted wait below relies on)
    dma_A(0);
    __builtin_amdgcn_sched_barrier(0);
    load_P(0, 0); load_P(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    for (int n = 0; n < NB_; ++n) {
        // RA(n) has landed (all but the NPL younger plain loads); every wave is done with block n-1's output products
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((AF_WI & 1) ? 0 : NPL) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(AF_WI & 2) || n == 0) dma_B(n);
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM 1: S[streamed token][own token]
        f32x16 s_[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_[mt][r] = 0.f;
#pragma unroll
            for (int ck = 0; ck < NCK; ++ck) {
                Frag fa;
                fa.hi = *reinterpret_cast<const s16x8a*>(RA + (ck * NBLK + 32 * mt) * RSZ + offh);
                fa.lo = *reinterpret_cast<const s16x8a*>(RA + (ck * NBLK + 32 * mt) * RSZ + offl);
                mma3(s_[mt], fa, y1[ck]);
                if (MODE == M_TAN) {
                    fa.hi = *reinterpret_cast<const s16x8a*>(RA + ABLK + (ck * NBLK + 32 * mt) * RSZ + offh);
                    fa.lo = *reinterpret_cast<const s16x8a*>(RA + ABLK + (ck * NBLK + 32 * mt) * RSZ + offl);
                    mma3(s_[mt], fa, y2[ck]);
                }
            }
        }
        // P(n) and RB(n) have landed.  (The scheduling fences pin the elementwise step between this wait and the barrier: left free,
        // the compiler sank it below the next DMA issue, where its own wait for the P registers became a vmcnt(0) on those DMAs.)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float w;
                if (MODE == M_TAN) { w = a.scale * pt[mt][r] * s_[mt][r]; rsum += w; }
                else if (MODE == M_COTQ) w = a.scale * pt[mt][r] * (s_[mt][r] - delta_own);
                else w = a.scale * pt[mt][r] * (s_[mt][r] - DL[wave * 64 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * khalf]);
                asm volatile("" : "+v"(w));           // (computed HERE: the optimiser otherwise sinks the step into the next phase's block)
                s_[mt][r] = w;
            }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // every wave is done with RA(n); RB(n) is visible
        asm volatile("" ::: "memory");
        const bool more = n + 1 < NB_;
        if (more && !(AF_WI & 2)) dma_A(n + 1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM 2: acc[channel][own token] += sum over the 64 streamed tokens; the P tile of block n+1 takes the registers
        // of each half as soon as that half's products are issued
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int jb = mt * 2 + bb;
                const Frag fw = (AF_WI & 4) ? y1[0] : frag_of(s_[mt], bb);
                Frag fp;
                if (MODE != M_COTQ) fp = (AF_WI & 4) ? y1[1] : frag_of(pt[mt], bb);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const unsigned char* r1 = RB + (jb * CHP + 32 * ct) * RSZ;
                    Frag f1, f2;
                    f1.hi = *reinterpret_cast<const s16x8a*>(r1 + offh);
                    f1.lo = *reinterpret_cast<const s16x8a*>(r1 + offl);
                    if (NTB == 2) {
                        f2.hi = *reinterpret_cast<const s16x8a*>(r1 + BBLK + offh);
                        f2.lo = *reinterpret_cast<const s16x8a*>(r1 + BBLK + offl);
                    }
                    if (MODE == M_TAN) {
                        mma3(acc[ct], f1, fw);                           // v W
                        mma3(acc[ct], f2, fp);                           // dv P
                    } else if (MODE == M_COTQ) {
                        mma3(acc[ct], f1, fw);                           // k G
                    } else {
                        mma3(acc[ct], f1, fp);                           // g_o P  -> g_v
                        mma3(acc2[ct], f2, fw);                          // q G    -> g_k
                    }
                }
            }
            if (more) load_P(n + 1, mt);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue.  D[row = channel][col = own token]
    if (MODE == M_TAN) rsum += __shfl_xor(rsum, 32, 64);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (c >= nch) continue;
            const long off = (long)c * T + mytok;
            if (MODE == M_TAN) {
                a.out[(long)b * a.bs_out + h * HO + off] = acc[ct][r] - rsum * o[off];
            } else if (MODE == M_COTQ) {
                a.gq[(long)b * a.bs_g + h * HS + off] = acc[ct][r];
            } else {
                a.gv[(long)b * a.bs_g + h * HS + off] = acc[ct][r];
                a.gk[(long)b * a.bs_g + h * HS + off] = acc2[ct][r];
            }
        }
}

}  // namespace

bool attn_flash_text_supported(int T, int CH, int Lt) {
    return CH >= 8 && CH <= 64 && T >= NOWN && (T % NOWN) == 0 && Lt > 0 && (Lt % NBLK) == 0;
}
bool attn_flash_supported(int T, int CH) {
    static int wide = -1;                 // LOCO_FLASH_WIDE=0: heads wider than 64 channels stay on the generic path (A/B switch)
    if (wide < 0) { const char* e = getenv("LOCO_FLASH_WIDE"); wide = e ? (atoi(e) != 0) : 1; }
    return CH >= 8 && CH <= (wide ? CHD : 64) && T >= NOWN && (T % NOWN) == 0;
}

template <int NCK, int NCT, bool TXT = false>
static void attn_flash_launch_n(int mode, const AttnFlashArgs& a, hipStream_t st) {
    dim3 grid(a.T / NOWN, a.NH, a.B);
    const size_t ldsb = 2 * region_bytes(NCT);       // NCT 2: 80 KB; NCT 3: 120 KB
    auto k0 = &attn_flash_kernel<M_TAN, NCK, NCT, TXT>;
    auto k1 = &attn_flash_kernel<M_COTQ, NCK, NCT, TXT>;
    auto k2 = &attn_flash_kernel<M_COTK, NCK, NCT, TXT>;
    static DeviceOnce once;
    if (first_on_device(once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    }
    if (mode == M_TAN) hipLaunchKernelGGL(k0, grid, dim3(256), ldsb, st, a);
    else if (mode == M_COTQ) hipLaunchKernelGGL(k1, grid, dim3(256), ldsb, st, a);
    else hipLaunchKernelGGL(k2, grid, dim3(256), ldsb, st, a);
}

// ---- the DMA-fed kernels: workspace layout, the split pass, the launches -------------------------------------------------
static int flash_dma_on() {          // LOCO_FLASH_DMA=0: the kernels that convert their operands themselves (A/B switch; read per launch)
    const char* e = getenv("LOCO_FLASH_DMA");
    return e ? atoi(e) : 1;
}
static void flash_shape(int CH, bool txt, int& nck, int& nct) {
    if (txt) { nck = 4; nct = 2; }
    else if (CH <= 48) { nck = 3; nct = 2; }
    else if (CH <= 64) { nck = 4; nct = 2; }
    else if (CH <= 80) { nck = 5; nct = 3; }
    else { nck = 6; nct = 3; }
}
struct FlashWs { size_t tokS, chS, tokT, chT, zero, tan, cot; };
static FlashWs flash_ws(const AttnFlashArgs& a) {
    int nck, nct;
    flash_shape(a.CH, a.Lt > 0, nck, nct);
    FlashWs w;
    const size_t ABLK = (size_t)nck * 4096, BBLK = (size_t)nct * 32 * 256;
    w.tokS = (size_t)a.NH * (a.T / NBLK) * ABLK;  w.chS = (size_t)a.NH * (a.T / NBLK) * BBLK;
    w.tokT = (size_t)a.NH * (a.Lt / NBLK) * ABLK; w.chT = (size_t)a.NH * (a.Lt / NBLK) * BBLK;
    w.zero = ABLK > BBLK ? ABLK : BBLK;
    w.tan = (size_t)(2 + 2 * a.B) * w.tokS + (size_t)(1 + a.B) * w.chS + w.tokT + w.chT + w.zero;
    w.cot = (size_t)(1 + a.B) * w.tokS + (size_t)(2 + a.B) * w.chS + w.tokT + w.chT + w.zero;
    return w;
}
size_t attn_flash_ws_bytes(const AttnFlashArgs& a) { const FlashWs w = flash_ws(a); return w.tan > w.cot ? w.tan : w.cot; }

static void flash_split(const SplitJobs& J, hipStream_t st) {
    int nb = 0, tmax = 0;
    for (int i = 0; i < J.n; ++i) { nb += J.j[i].nb; tmax = J.j[i].T > tmax ? J.j[i].T : tmax; }
    hipLaunchKernelGGL(attn_split_kernel, dim3(tmax / NBLK, J.NH, nb), dim3(256), 0, st, J);
}

template <int MODE, int NCK, int NCT, bool TXT>
static void flash_dma_launch_one(const AttnFlashArgs& a, const FlashRecs& R, hipStream_t st) {
    constexpr int NTA = MODE == M_TAN ? 2 : 1, NTB = MODE == M_COTQ ? 1 : 2;
    const size_t ldsb = (size_t)NTA * NCK * 4096 + (size_t)NTB * NCT * 32 * 256 + 1024;
    auto k = &attn_flash_dma_kernel<MODE, NCK, NCT, TXT>;
    static DeviceOnce once;
    if (first_on_device(once)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    if (AF_WI & 8) return;
    hipLaunchKernelGGL(k, dim3((a.T / NOWN) * a.NH * a.B), dim3(256), ldsb, st, a, R);
}

template <int NCK, int NCT, bool TXT = false>
static void attn_flash_dma_n(int mode, const AttnFlashArgs& a, hipStream_t st) {      // mode: M_TAN, or M_COTQ for the cotangent pair
    const FlashWs w = flash_ws(a);
    unsigned char* p = a.ws;
    auto take = [&](size_t n) { unsigned char* r = p; p += n; return r; };
    SplitJobs J; std::memset(&J, 0, sizeof(J));
    J.NCK = NCK; J.CHP = 32 * NCT; J.nch = a.CH; J.NH = a.NH;
    auto job = [&](const float* src, long bs, long hs, int nb, int T, int layout, unsigned char* dst) {
        J.j[J.n++] = SplitJob{src, bs, hs, nb, T, layout, dst};
    };
    const long HO = (long)a.CH * a.T;
    FlashRecs R; std::memset(&R, 0, sizeof(R));
    unsigned char* zero = take(w.zero);
    (void)hipMemsetAsync(zero, 0, w.zero, st);
    R.zero = zero;
    if (mode == M_TAN) {
        unsigned char *rk = take(w.tokS), *rdk = take(a.B * w.tokS), *rq = take(w.tokS), *rdq = take(a.B * w.tokS);
        unsigned char *rv = take(w.chS), *rdv = take(a.B * w.chS);
        job(a.k, 0, a.hs, 1, a.T, 0, rk);   job(a.dk, a.bs_d, a.hs, a.B, a.T, 0, rdk);
        job(a.q, 0, a.hs, 1, a.T, 0, rq);   job(a.dq, a.bs_d, a.hs, a.B, a.T, 0, rdq);
        job(a.v, 0, a.hs, 1, a.T, 1, rv);   job(a.dv, a.bs_d, a.hs, a.B, a.T, 1, rdv);
        if (TXT) {
            unsigned char *rkt = take(w.tokT), *rvt = take(w.chT);
            job(a.kt, 0, (long)a.CH * a.Lt, 1, a.Lt, 0, rkt); job(a.vt, 0, (long)a.CH * a.Lt, 1, a.Lt, 1, rvt);
            R.ta1 = rkt; R.tb1 = rvt;
        }
        flash_split(J, st);
        R.own1 = rdq; R.own1_bs = (long)w.tokS; R.own2 = rq; R.own2_bs = 0;
        R.sa1 = rk; R.sa1_bs = 0; R.sa2 = rdk; R.sa2_bs = (long)w.tokS;
        R.sb1 = rv; R.sb1_bs = 0; R.sb2 = rdv; R.sb2_bs = (long)w.chS;
        flash_dma_launch_one<M_TAN, NCK, NCT, TXT>(a, R, st);
        return;
    }
    unsigned char *rgo = take(a.B * w.tokS), *rv = take(w.tokS), *rk = take(w.chS), *rgoc = take(a.B * w.chS), *rq = take(w.chS);
    job(a.go, a.bs_go, HO, a.B, a.T, 0, rgo);  job(a.v, 0, a.hs, 1, a.T, 0, rv);
    job(a.k, 0, a.hs, 1, a.T, 1, rk);          job(a.go, a.bs_go, HO, a.B, a.T, 1, rgoc);
    job(a.q, 0, a.hs, 1, a.T, 1, rq);
    if (TXT) {
        unsigned char *rvt = take(w.tokT), *rkt = take(w.chT);
        job(a.vt, 0, (long)a.CH * a.Lt, 1, a.Lt, 0, rvt); job(a.kt, 0, (long)a.CH * a.Lt, 1, a.Lt, 1, rkt);
        R.ta1 = rvt; R.tb1 = rkt;
    }
    flash_split(J, st);
    R.own1 = rgo; R.own1_bs = (long)w.tokS;
    R.sa1 = rv; R.sa1_bs = 0;
    R.sb1 = rk; R.sb1_bs = 0;
    flash_dma_launch_one<M_COTQ, NCK, NCT, TXT>(a, R, st);      // g_q and delta_i (read by the second kernel: the kernel boundary orders them)
    FlashRecs K = R;
    K.own1 = rv; K.own1_bs = 0;
    K.sa1 = rgo; K.sa1_bs = (long)w.tokS;
    K.sb1 = rgoc; K.sb1_bs = (long)w.chS; K.sb2 = rq; K.sb2_bs = 0;
    flash_dma_launch_one<M_COTK, NCK, NCT, TXT>(a, K, st);      // g_k, g_v
}

static void attn_flash_launch(int mode, const AttnFlashArgs& a, hipStream_t st) {
    if (a.ws && flash_dma_on() && attn_flash_ws_bytes(a) <= a.ws_bytes && (mode == M_TAN || mode == M_COTQ)) {
        if (a.Lt > 0) attn_flash_dma_n<4, 2, true>(mode, a, st);
        else if (a.CH <= 48) attn_flash_dma_n<3, 2>(mode, a, st);
        else if (a.CH <= 64) attn_flash_dma_n<4, 2>(mode, a, st);
        else if (a.CH <= 80) attn_flash_dma_n<5, 3>(mode, a, st);
        else attn_flash_dma_n<6, 3>(mode, a, st);
        return;
    }
    if (a.Lt > 0) { attn_flash_launch_n<4, 2, true>(mode, a, st); return; }      // attn_flash_text_supported: heads of <= 64 channels
    if (a.CH <= 48) attn_flash_launch_n<3, 2>(mode, a, st);
    else if (a.CH <= 64) attn_flash_launch_n<4, 2>(mode, a, st);
    else if (a.CH <= 80) attn_flash_launch_n<5, 3>(mode, a, st);
    else attn_flash_launch_n<6, 3>(mode, a, st);
}

void launch_attn_flash_tangent(const AttnFlashArgs& a, hipStream_t st) { attn_flash_launch(M_TAN, a, st); }
void launch_attn_flash_cotangent(const AttnFlashArgs& a, hipStream_t st) {
    if (a.ws && flash_dma_on() && attn_flash_ws_bytes(a) <= a.ws_bytes) { attn_flash_launch(M_COTQ, a, st); return; }    // (the pair)
    attn_flash_launch(M_COTQ, a, st);      // g_q and delta_i (read by the second kernel: the kernel boundary orders them)
    attn_flash_launch(M_COTK, a, st);      // g_k, g_v
}

}  // namespace loco
