// Dense algebra of the block power iteration on tall-skinny k x n row blocks
// (k <= 64, n = C*H*W): Gram in double, one-workgroup parallel Jacobi
// eigensolver, row rotation V = S^-1 Q A (the thin SVD's Vh), CholeskyQR,
// convergence norm, null-space projection, edit axpy, masked gather.
// Replaces torch.linalg.qr / torch.linalg.svd / torch.dist / allclose on the
// host round trips of reference edit.py:2435-2436, 2482, 2489-2492, 2317-2323.
#include "kernels.h"

namespace loco {

constexpr int GCH = 256;   // columns per Gram block

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// partial cross Gram of a column chunk: P[blk][i][j] = sum_{c in chunk} A[i][c] * B[j][c]
__global__ __launch_bounds__(256) void cross_gram_partial(const float* A, int k1, const float* B, int k2, long n,
                                                          double* part) {
    extern __shared__ float sm[];            // [k1][GCH+1] + [k2][GCH+1]
    float* As = sm;
    float* Bs = sm + (long)k1 * (GCH + 1);
    const long c0 = (long)blockIdx.x * GCH;
    for (int e = threadIdx.x; e < k1 * GCH; e += 256) {
        int i = e / GCH, c = e % GCH;
        As[i * (GCH + 1) + c] = (c0 + c < n) ? A[(long)i * n + c0 + c] : 0.f;
    }
    if (B != A) {
        for (int e = threadIdx.x; e < k2 * GCH; e += 256) {
            int i = e / GCH, c = e % GCH;
            Bs[i * (GCH + 1) + c] = (c0 + c < n) ? B[(long)i * n + c0 + c] : 0.f;
        }
    } else {
        Bs = As;
    }
    __syncthreads();
    // one wave per (i,j) pair, lanes stride the chunk
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int pr = wave; pr < k1 * k2; pr += 4) {
        int i = pr / k2, j = pr % k2;
        double acc = 0.0;
#pragma unroll
        for (int c = lane; c < GCH; c += 64)
            acc += (double)As[i * (GCH + 1) + c] * (double)Bs[j * (GCH + 1) + c];
        acc = wsum(acc);
        if (lane == 0) part[(long)blockIdx.x * k1 * k2 + pr] = acc;
    }
}
// one workgroup per Gram entry: fixed-order strided partial sums, then an LDS tree (deterministic)
__global__ __launch_bounds__(256) void gram_reduce(const double* part, int nblk, int kk, double* G) {
    __shared__ double red[256];
    const int p = blockIdx.x;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) acc += part[(long)b * kk + p];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) G[p] = red[0];
}
void launch_cross_gram(const float* A, int k1, const float* B, int k2, long n, double* C, double* scratch,
                       hipStream_t st) {
    int nblk = (int)((n + GCH - 1) / GCH);
    size_t lds = (size_t)(k1 + (B != A ? k2 : 0)) * (GCH + 1) * sizeof(float);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_gram_partial),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cross_gram_partial, dim3(nblk), dim3(256), lds, st, A, k1, B, k2, n, scratch);
    hipLaunchKernelGGL(gram_reduce, dim3(k1 * k2), dim3(256), 0, st, scratch, nblk, k1 * k2, C);
}
void launch_gram(const float* A, int k, long n, double* G, double* scratch, hipStream_t st) {
    launch_cross_gram(A, k, A, k, n, G, scratch, st);
}

// ---------------------------------------------------------------------------
// Parallel two-sided Jacobi for a symmetric k x k matrix in double (k <= 64).
// Round-robin ordering: kp = k rounded up to even, kp-1 rounds per sweep, kp/2
// disjoint rotations per round.  Output: w descending, Q rows = eigenvectors.
__global__ __launch_bounds__(256) void jacobi_eig_kernel(double* Gio, int k, double* w, double* Q) {
    __shared__ double G[64][65];
    __shared__ double V[64][65];
    __shared__ double cs[32], sn[32];
    __shared__ int pp[32], qq[32];
    __shared__ int order[64];
    __shared__ double red_off[256], red_dia[256];
    const int tid = threadIdx.x;
    const int kp = (k + 1) & ~1;
    for (int e = tid; e < 64 * 64; e += 256) {
        int i = e >> 6, j = e & 63;
        G[i][j] = (i < k && j < k) ? Gio[i * k + j] : 0.0;
        V[i][j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    const int npair = kp / 2;
    for (int sweep = 0; sweep < 16; ++sweep) {
        for (int round = 0; round < kp - 1; ++round) {
            if (tid < npair) {
                // round-robin tournament on m = kp-1 rotating players + one fixed (index m)
                const int m = kp - 1;
                int a, b;
                if (tid == 0) { a = m; b = round; }
                else { a = (round + tid) % m; b = (round - tid + m) % m; }
                int p = a < b ? a : b, q = a < b ? b : a;
                double c = 1.0, s = 0.0;
                if (q < k) {
                    double apq = G[p][q];
                    if (fabs(apq) > 1e-300) {
                        double tau = (G[q][q] - G[p][p]) / (2.0 * apq);
                        double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                }
                pp[tid] = p; qq[tid] = q; cs[tid] = c; sn[tid] = s;
            }
            __syncthreads();
            // rows: G <- J^T G
            for (int e = tid; e < npair * k; e += 256) {
                int pr = e / k, j = e % k;
                int p = pp[pr], q = qq[pr];
                if (q < k) {
                    double c = cs[pr], s = sn[pr];
                    double gp = G[p][j], gq = G[q][j];
                    G[p][j] = c * gp - s * gq;
                    G[q][j] = s * gp + c * gq;
                }
            }
            __syncthreads();
            // columns: G <- G J ; V <- V J
            for (int e = tid; e < npair * k; e += 256) {
                int pr = e / k, j = e % k;
                int p = pp[pr], q = qq[pr];
                if (q < k) {
                    double c = cs[pr], s = sn[pr];
                    double gp = G[j][p], gq = G[j][q];
                    G[j][p] = c * gp - s * gq;
                    G[j][q] = s * gp + c * gq;
                    double vp = V[j][p], vq = V[j][q];
                    V[j][p] = c * vp - s * vq;
                    V[j][q] = s * vp + c * vq;
                }
            }
            __syncthreads();
        }
        // converged?  max |off-diagonal| against max |diagonal| (cyclic Jacobi converges quadratically: 5-8 sweeps)
        double off = 0.0, dia = 0.0;
        for (int e = tid; e < k * k; e += 256) {
            int i = e / k, j = e % k;
            double v = fabs(G[i][j]);
            if (i == j) dia = v > dia ? v : dia; else off = v > off ? v : off;
        }
        red_off[tid] = off; red_dia[tid] = dia;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                if (red_off[tid + st] > red_off[tid]) red_off[tid] = red_off[tid + st];
                if (red_dia[tid + st] > red_dia[tid]) red_dia[tid] = red_dia[tid + st];
            }
            __syncthreads();
        }
        const bool done = red_off[0] <= 1e-15 * red_dia[0];
        __syncthreads();
        if (done) break;
    }
    if (tid == 0) {
        for (int i = 0; i < k; ++i) order[i] = i;
        for (int i = 0; i < k; ++i) {
            int best = i;
            for (int j = i + 1; j < k; ++j)
                if (G[order[j]][order[j]] > G[order[best]][order[best]]) best = j;
            int t = order[i]; order[i] = order[best]; order[best] = t;
        }
    }
    __syncthreads();
    for (int e = tid; e < k * k; e += 256) {
        int i = e / k, j = e % k;
        Q[e] = V[j][order[i]];          // row i = eigenvector of the i-th largest eigenvalue
    }
    if (tid < k) w[tid] = G[order[tid]][order[tid]];
}
void launch_jacobi_eig(double* G, int k, double* w, double* Q, hipStream_t st) {
    hipLaunchKernelGGL(jacobi_eig_kernel, dim3(1), dim3(256), 0, st, G, k, w, Q);
}

// Aout[i][c] = scale_i * sum_j Q[i][j] * Ain[j][c]
__global__ __launch_bounds__(256) void rotate_rows_kernel(const float* Ain, float* Aout, int k, long n,
                                                          const double* Q, const double* w, int mode) {
    extern __shared__ double qs[];      // [k][k] scaled
    for (int e = threadIdx.x; e < k * k; e += 256) {
        int i = e / k;
        double sc = 1.0;
        if (mode == 0) {
            double ev = w[i];
            sc = ev > 1e-300 ? 1.0 / sqrt(ev) : 0.0;
        }
        qs[e] = Q[e] * sc;
    }
    __syncthreads();
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n; c += (long)gridDim.x * 256) {
        float col[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) col[j] = (j < k) ? Ain[(long)j * n + c] : 0.f;
        for (int i = 0; i < k; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 64; ++j)
                if (j < k) acc += qs[i * k + j] * (double)col[j];
            Aout[(long)i * n + c] = (float)acc;
        }
    }
}
void launch_rotate_rows(const float* Ain, float* Aout, int k, long n, const double* Q, const double* w, int mode,
                        hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(rotate_rows_kernel, dim3(blocks), dim3(256), (size_t)k * k * sizeof(double), st, Ain, Aout,
                       k, n, Q, w, mode);
}

// per row: flip so that the entry of largest magnitude is positive; also emit s = sqrt(max(w,0)).
// Two passes over SEG-element segments so the whole chip works on the k rows: candidates, then decide + flip.
constexpr int SIGN_SEG = 4096;
__global__ __launch_bounds__(256) void sign_cand_kernel(const float* A, long n, int nseg, float* cand) {
    __shared__ float bestv[256];
    const float* row = A + (long)blockIdx.y * n;
    const long c0 = (long)blockIdx.x * SIGN_SEG;
    float bv = 0.f;
    for (int c = threadIdx.x; c < SIGN_SEG && c0 + c < n; c += 256) {
        float v = row[c0 + c];
        if (fabsf(v) > fabsf(bv)) bv = v;
    }
    bestv[threadIdx.x] = bv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s && fabsf(bestv[threadIdx.x + s]) > fabsf(bestv[threadIdx.x]))
            bestv[threadIdx.x] = bestv[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) cand[(long)blockIdx.y * nseg + blockIdx.x] = bestv[0];
}
__global__ __launch_bounds__(256) void sign_apply_kernel(float* A, long n, int nseg, const float* cand, float* s_out,
                                                         const double* w) {
    __shared__ float bestv[256];
    float bv = 0.f;
    for (int i = threadIdx.x; i < nseg; i += 256) {
        float v = cand[(long)blockIdx.y * nseg + i];
        if (fabsf(v) > fabsf(bv)) bv = v;
    }
    bestv[threadIdx.x] = bv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s && fabsf(bestv[threadIdx.x + s]) > fabsf(bestv[threadIdx.x]))
            bestv[threadIdx.x] = bestv[threadIdx.x + s];
        __syncthreads();
    }
    if (bestv[0] < 0.f) {
        float* row = A + (long)blockIdx.y * n;
        const long c0 = (long)blockIdx.x * SIGN_SEG;
        for (int c = threadIdx.x; c < SIGN_SEG && c0 + c < n; c += 256) row[c0 + c] = -row[c0 + c];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && s_out) {
        double ev = w[blockIdx.y];
        s_out[blockIdx.y] = (float)sqrt(ev > 0 ? ev : 0.0);
    }
}
void launch_sign_fix(float* A, int k, long n, float* s_out, const double* w, float* scratch, hipStream_t st) {
    const int nseg = (int)((n + SIGN_SEG - 1) / SIGN_SEG);
    hipLaunchKernelGGL(sign_cand_kernel, dim3(nseg, k), dim3(256), 0, st, A, n, nseg, scratch);
    hipLaunchKernelGGL(sign_apply_kernel, dim3(nseg, k), dim3(256), 0, st, A, n, nseg, scratch, s_out, w);
}

// in-place lower Cholesky of a k x k double matrix (row-major), one workgroup
__global__ __launch_bounds__(64) void cholesky_kernel(double* G, int k) {
    __shared__ double L[64][65];
    const int t = threadIdx.x;
    for (int e = t; e < k * k; e += 64) L[e / k][e % k] = G[e];
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        if (t == 0) {
            double d = L[j][j];
            for (int p = 0; p < j; ++p) d -= L[j][p] * L[j][p];
            L[j][j] = sqrt(d > 1e-300 ? d : 1e-300);
        }
        __syncthreads();
        if (t > j && t < k) {
            double v = L[t][j];
            for (int p = 0; p < j; ++p) v -= L[t][p] * L[j][p];
            L[t][j] = v / L[j][j];
        }
        __syncthreads();
    }
    for (int e = t; e < k * k; e += 64) {
        int i = e / k, j = e % k;
        G[e] = (j <= i) ? L[i][j] : 0.0;
    }
}
void launch_cholesky(double* G, int k, hipStream_t st) {
    hipLaunchKernelGGL(cholesky_kernel, dim3(1), dim3(64), 0, st, G, k);
}
// Aout = L^{-1} Ain (forward substitution per column)
__global__ __launch_bounds__(256) void trsm_rows_kernel(const float* Ain, float* Aout, int k, long n,
                                                        const double* L) {
    extern __shared__ double ls[];
    for (int e = threadIdx.x; e < k * k; e += 256) ls[e] = L[e];
    __syncthreads();
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n; c += (long)gridDim.x * 256) {
        double y[64];
        for (int i = 0; i < k; ++i) {
            double v = (double)Ain[(long)i * n + c];
            for (int j = 0; j < i; ++j) v -= ls[i * k + j] * y[j];
            y[i] = v / ls[i * k + i];
            Aout[(long)i * n + c] = (float)y[i];
        }
    }
}
void launch_trsm_rows(const float* Ain, float* Aout, int k, long n, const double* L, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(trsm_rows_kernel, dim3(blocks), dim3(256), (size_t)k * k * sizeof(double), st, Ain, Aout, k,
                       n, L);
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_partial_kernel(const float* a, const float* b, long count, float atol,
                                                           float rtol, double* part) {
    __shared__ double sm[4];
    __shared__ int bad[4];
    double ss = 0.0;
    int nb = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
        float d = a[i] - b[i];
        ss += (double)d * (double)d;
        if (!(fabsf(d) <= atol + rtol * fabsf(b[i]))) nb = 1;
    }
    ss = wsum(ss);
    nb = __any(nb) ? 1 : 0;
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = ss; bad[threadIdx.x >> 6] = nb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
        part[2 * blockIdx.x + 1] = (double)(bad[0] | bad[1] | bad[2] | bad[3]);
    }
}
__global__ void conv_final_kernel(const double* part, int nblk, float* out2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double ss = 0.0, bad = 0.0;
        for (int i = 0; i < nblk; ++i) { ss += part[2 * i]; bad += part[2 * i + 1]; }
        out2[0] = (float)sqrt(ss);
        out2[1] = bad > 0 ? 0.f : 1.f;
    }
}
void launch_convergence(const float* a, const float* b, long count, float atol, float rtol, float* out2,
                        double* scratch, hipStream_t st) {
    int blocks = (int)((count + 255) / 256);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(conv_partial_kernel, dim3(blocks), dim3(256), 0, st, a, b, count, atol, rtol, scratch);
    hipLaunchKernelGGL(conv_final_kernel, dim3(1), dim3(64), 0, st, scratch, blocks, out2);
}

// The same test per ROW and up to the row's sign (edit.py:2489-2492 compares LAPACK's singular vectors, whose signs are
// LAPACK's choice; the eigenvectors here carry no sign of their own): for every row both orientations are accumulated
// in one pass -- ||a - b||^2, ||a + b||^2 and the allclose verdict of each -- and the final kernel keeps, per row, the
// orientation with the smaller distance.  out2[0] = sqrt(sum_rows min(||a-b||^2, ||a+b||^2)), out2[1] = 1 when every
// row is allclose in its orientation.  grid (segments, k); part: [k][nseg][4] doubles.
__global__ __launch_bounds__(256) void conv_rows_partial_kernel(const float* a, const float* b, long n, float atol,
                                                                float rtol, double* part) {
    __shared__ double sp[4], sm_[4];
    __shared__ int bp[4], bm[4];
    const long r0 = (long)blockIdx.y * n;
    double ssp = 0.0, ssm = 0.0;
    int nbp = 0, nbm = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = a[r0 + i], y = b[r0 + i];
        const float dp = x - y, dm = x + y;
        ssp += (double)dp * (double)dp;
        ssm += (double)dm * (double)dm;
        const float tol = atol + rtol * fabsf(y);
        if (!(fabsf(dp) <= tol)) nbp = 1;
        if (!(fabsf(dm) <= tol)) nbm = 1;
    }
    ssp = wsum(ssp); ssm = wsum(ssm);
    nbp = __any(nbp) ? 1 : 0; nbm = __any(nbm) ? 1 : 0;
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; sp[w] = ssp; sm_[w] = ssm; bp[w] = nbp; bm[w] = nbm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* o = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        o[0] = sp[0] + sp[1] + sp[2] + sp[3];
        o[1] = sm_[0] + sm_[1] + sm_[2] + sm_[3];
        o[2] = (double)(bp[0] | bp[1] | bp[2] | bp[3]);
        o[3] = (double)(bm[0] | bm[1] | bm[2] | bm[3]);
    }
}
__global__ void conv_rows_final_kernel(const double* part, int nseg, int k, float* out2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double ss = 0.0;
        int bad = 0;
        for (int r = 0; r < k; ++r) {
            double p = 0.0, m = 0.0, badp = 0.0, badm = 0.0;
            for (int i = 0; i < nseg; ++i) {
                const double* o = part + ((long)r * nseg + i) * 4;
                p += o[0]; m += o[1]; badp += o[2]; badm += o[3];
            }
            if (m < p) { ss += m; bad |= badm > 0; } else { ss += p; bad |= badp > 0; }
        }
        out2[0] = (float)sqrt(ss);
        out2[1] = bad ? 0.f : 1.f;
    }
}
void launch_convergence_rows(const float* a, const float* b, int k, long n, float atol, float rtol, float* out2,
                             double* scratch, hipStream_t st) {
    int nseg = (int)((n + 255) / 256);
    if (nseg > 64) nseg = 64;
    hipLaunchKernelGGL(conv_rows_partial_kernel, dim3(nseg, k), dim3(256), 0, st, a, b, n, atol, rtol, scratch);
    hipLaunchKernelGGL(conv_rows_final_kernel, dim3(1), dim3(64), 0, st, scratch, nseg, k, out2);
}

// out[i][c] = Vm[i][c] - sum_j C[j][i] * Vn[j][c]        (C = Vn Vm^T, [k0][k])
__global__ __launch_bounds__(256) void project_rows_kernel(const float* Vm, int k, const float* Vn, int k0, long n,
                                                           const double* C, float* out) {
    extern __shared__ double cs[];
    for (int e = threadIdx.x; e < k0 * k; e += 256) cs[e] = C[e];
    __syncthreads();
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n; c += (long)gridDim.x * 256) {
        float vn[64];
#pragma unroll
        for (int j = 0; j < 64; ++j) vn[j] = (j < k0) ? Vn[(long)j * n + c] : 0.f;
        for (int i = 0; i < k; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 64; ++j)
                if (j < k0) acc += cs[j * k + i] * (double)vn[j];
            out[(long)i * n + c] = (float)((double)Vm[(long)i * n + c] - acc);
        }
    }
}
void launch_project_rows(const float* Vm, int k, const float* Vn, int k0, long n, const double* C, float* out,
                         hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(project_rows_kernel, dim3(blocks), dim3(256), (size_t)k0 * k * sizeof(double), st, Vm, k,
                       Vn, k0, n, C, out);
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(float* A, long n) {
    __shared__ double sm[4];
    float* row = A + (long)blockIdx.x * n;
    double ss = 0.0;
    for (long c = threadIdx.x; c < n; c += 256) ss += (double)row[c] * (double)row[c];
    ss = wsum(ss);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
    __syncthreads();
    float nrm = (float)sqrt(sm[0] + sm[1] + sm[2] + sm[3]);
    for (long c = threadIdx.x; c < n; c += 256) row[c] = row[c] / nrm;
}
void launch_normalize_rows(float* A, int k, long n, double* scratch, hipStream_t st) {
    (void)scratch;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(k), dim3(256), 0, st, A, n);
}

__global__ void edit_axpy_kernel(const float* x, const float* v, const float* alphas, long n, float* out) {
    const int b = blockIdx.y;
    const float al = alphas[b];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[(long)b * n + i] = x[i] + al * v[i];
}
void launch_edit_axpy(const float* x, const float* v, const float* alphas_dev, int B, long n, float* out,
                      hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(edit_axpy_kernel, dim3(blocks, B), dim3(256), 0, st, x, v, alphas_dev, n, out);
}

__global__ void mask_gather_kernel(const float* U, const int* idx, long L, long n, float* out) {
    const int b = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x)
        out[(long)b * L + i] = U[(long)b * n + idx[i]];
}
void launch_mask_gather(const float* U, const int* idx, long L, long n, int k, float* out, hipStream_t st) {
    int blocks = (int)((L + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mask_gather_kernel, dim3(blocks, k), dim3(256), 0, st, U, idx, L, n, out);
}

// Ordered stream compaction of the mask (north_star: coalesced masked-latent gather): one workgroup of 1024 threads;
// thread t owns the contiguous slice [t*per, (t+1)*per) so the output order is ascending; counts are combined with a
// wave-shuffle inclusive scan + one LDS pass over the 16 wave totals.
__global__ __launch_bounds__(1024) void mask_compact_kernel(const uint8_t* mask, long n, int* idx, int* count) {
    __shared__ int wtot[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long per = (n + 1023) / 1024;
    const long lo = (long)t * per, hi = lo + per < n ? lo + per : n;
    int cnt = 0;
    for (long i = lo; i < hi; ++i) cnt += mask[i] != 0;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
    }
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wtot[w];
    int pos = base + inc - cnt;
    for (long i = lo; i < hi; ++i)
        if (mask[i]) idx[pos++] = (int)i;
    if (t == 1023) *count = base + inc;
}
void launch_mask_compact(const uint8_t* mask, long n, int* idx, int* count, hipStream_t st) {
    hipLaunchKernelGGL(mask_compact_kernel, dim3(1), dim3(1024), 0, st, mask, n, idx, count);
}

}  // namespace loco
