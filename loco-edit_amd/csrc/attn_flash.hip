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

__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
    unsigned h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 hb = (__bf16)v[j];
        __bf16 lb = (__bf16)(v[j] - (float)hb);
        h[j] = __builtin_bit_cast(unsigned short, hb);
        l[j] = __builtin_bit_cast(unsigned short, lb);
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
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
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
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
template <int MODE, int NCK, int NCT, bool TXT = false>
__global__ __launch_bounds__(256) void attn_flash_kernel(AttnFlashArgs a) {
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
        store_tokens<NBLK, NOCT>(ta, RA, tid);
        store_channels<NCP, CHP>(ca, RB, tid);
        if (MODE == M_TAN) { store_tokens<NBLK, NOCT>(tb, RA + T2, tid); store_channels<NCP, CHP>(cb, RB + T2, tid); }
        if (MODE == M_COTK) { store_channels<NCP, CHP>(cb, RB + T2, tid); if (tid < NBLK) DL[tid] = dreg; }
        if (t0 + NBLK < U) fetch(t0 + NBLK);        // in flight under this block's MFMAs
        // primal probabilities of the tile in the D-fragment layout (rows = streamed tokens, column = own token)
        f32x16 pt[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
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
                const Frag fw = frag_of(s_[mt], bb);
                Frag fp;
                if (MODE != M_COTQ) fp = frag_of(pt[mt], bb);
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
    const size_t ldsb = 2 * region_bytes(NCT);       // NCT 2: 80 KB, two workgroups per CU; NCT 3: 120 KB, one
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
static void attn_flash_launch(int mode, const AttnFlashArgs& a, hipStream_t st) {
    if (a.Lt > 0) { attn_flash_launch_n<4, 2, true>(mode, a, st); return; }      // attn_flash_text_supported: heads of <= 64 channels
    if (a.CH <= 48) attn_flash_launch_n<3, 2>(mode, a, st);
    else if (a.CH <= 64) attn_flash_launch_n<4, 2>(mode, a, st);
    else if (a.CH <= 80) attn_flash_launch_n<5, 3>(mode, a, st);
    else attn_flash_launch_n<6, 3>(mode, a, st);
}

void launch_attn_flash_tangent(const AttnFlashArgs& a, hipStream_t st) { attn_flash_launch(M_TAN, a, st); }
void launch_attn_flash_cotangent(const AttnFlashArgs& a, hipStream_t st) {
    attn_flash_launch(M_COTQ, a, st);      // g_q and delta_i (read by the second kernel: the kernel boundary orders them)
    attn_flash_launch(M_COTK, a, st);      // g_k, g_v
}

}  // namespace loco
