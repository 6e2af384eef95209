// VERDICT r3 next #5 -- "decide MFMA on evidence".  Standalone prototype of the screen-then-exact-rescore scheme on the shape of the
// dense-768 LEAF (BASELINE.json configs[4]: D = 768 dense fp32 queries, ~92-column leaf chunks, every query visits 10 beam parents):
//
//   1. PRE-SCORE   s'[item][col] = bias[col] + sum_k x[q][k] * w[k][col] with v_mfma_f32_16x16x4_f32 -- an fp32 FMA chain (ONE rounding per
//                  step), which is NOT the reference's arithmetic (multiply and add rounded separately, inference.hpp:512-517, 823-837).
//   2. BAND        per query, a candidate can only belong to the exact top-k if  s' + b  >=  the k-th largest of (s' - b),  b = a proven bound
//                  on |s' - s|:  both chains approximate the real sum, the two-rounding chain within gamma_{2D+1} * sum|x w| and the FMA chain
//                  within gamma_{D+1} * sum|x w|, so b = (3D + 2) u * sum_k |x_k w_k|  (u = 2^-24); cheaper to obtain is the Cauchy-Schwarz
//                  form  b_cs = (3D + 2) u * ||x|| * ||w_col||  (column norms at load, row norms once per batch).
//   3. RESCORE     only the candidates inside the band run the reference's sequential arithmetic (what K1G does for ALL of them today).
//
// This program measures: the MFMA pre-score's rate on tile-sorted (query, parent) items against the dense row format; the observed
// |s' - s| against both bounds; the fraction of candidates a top-10 selection has to re-score; and the cost of re-scoring them with one lane
// per candidate.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/mfma_prescore.hip -o scripts/bin/mfma_prescore
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int D = 768, C = 96, BEAM = 10, QB = 64, KC = 32, XSP = 34, WSP = 112;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Item { uint32_t q, tile; };

// one workgroup = up to 64 tile-sorted items of ONE tile x its 96 columns; 4 wavefronts x (16 items x 6 column blocks of 16)
__global__ void __launch_bounds__(256) prescore_mfma(const float* __restrict__ X, const float* __restrict__ W, uint64_t ld, const float* __restrict__ bias,
                                                     const Item* __restrict__ items, const uint32_t* __restrict__ blk_item0, const uint32_t* __restrict__ blk_n,
                                                     float* __restrict__ out) {
    __shared__ float xs[2][QB * XSP];
    __shared__ float ws[2][KC * WSP];
    const uint32_t i0 = blk_item0[blockIdx.x], n = blk_n[blockIdx.x];
    const uint32_t tile = items[i0].tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // staging assignments: X panel 64 rows x 32 floats = 512 float4 (2 per thread), W panel 32 rows x 96 floats = 768 float4 (3 per thread)
    uint32_t xq[2]; int xr[2], xc[2];
    for (int j = 0; j < 2; ++j) { const int v = tid + 256 * j; xr[j] = v >> 3; xc[j] = (v & 7) * 4; xq[j] = items[i0 + min((uint32_t)xr[j], n - 1)].q; }
    int wr[3], wc[3];
    for (int j = 0; j < 3; ++j) { const int v = tid + 256 * j; wr[j] = v / 24; wc[j] = (v % 24) * 4; }
    const float* __restrict__ Wt = W + (uint64_t)tile * C;
    float4 xg[2], wg[3];
    auto gload = [&](int k0) {
        for (int j = 0; j < 2; ++j) xg[j] = *reinterpret_cast<const float4*>(X + (uint64_t)xq[j] * D + k0 + xc[j]);
        for (int j = 0; j < 3; ++j) wg[j] = *reinterpret_cast<const float4*>(Wt + (uint64_t)(k0 + wr[j]) * ld + wc[j]);
    };
    auto sstore = [&](int b) {
        for (int j = 0; j < 2; ++j) { float* p = &xs[b][xr[j] * XSP + xc[j]]; p[0] = xg[j].x; p[1] = xg[j].y; p[2] = xg[j].z; p[3] = xg[j].w; }
        for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(&ws[b][wr[j] * WSP + wc[j]]) = wg[j];
    };
    f32x4 acc[6];
    const int col = lane & 15, kq = lane >> 4;
    for (int j = 0; j < 6; ++j) { const float b = bias[(uint64_t)tile * C + j * 16 + col]; acc[j] = f32x4{b, b, b, b}; }
    gload(0); sstore(0);
    __syncthreads();
    for (int c = 0; c < D / KC; ++c) {
        const int b = c & 1;
        if (c + 1 < D / KC) gload((c + 1) * KC);
        const float* __restrict__ xa = &xs[b][(wave * 16 + col) * XSP + kq];
        const float* __restrict__ wa = &ws[b][kq * WSP + col];
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {
            const float a = xa[kk * 4];
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wa[kk * 4 * WSP + j * 16], acc[j], 0, 0, 0);
        }
        if (c + 1 < D / KC) sstore(b ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
    for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = (uint32_t)(wave * 16 + kq * 4 + r);
            if (row < n) out[(uint64_t)(i0 + row) * C + j * 16 + col] = acc[j][r];
        }
}

// the reference's arithmetic for EVERY candidate (one thread per (item, column), columns adjacent: coalesced weight reads) + sum |x w|
__global__ void __launch_bounds__(256) exact_all(const float* __restrict__ X, const float* __restrict__ W, uint64_t ld, const float* __restrict__ bias,
                                                 const Item* __restrict__ items, uint64_t n_items, float* __restrict__ out, float* __restrict__ out_abs) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_items * C) return;
    const uint64_t it = t / C; const uint32_t col = (uint32_t)(t % C);
    const Item I = items[it];
    const float* __restrict__ x = X + (uint64_t)I.q * D;
    const float* __restrict__ w = W + (uint64_t)I.tile * C + col;
    float s = bias[(uint64_t)I.tile * C + col], sa = fabsf(s);
    for (int k = 0; k < D; ++k) { const float p = __fmul_rn(x[k], w[(uint64_t)k * ld]); s = __fadd_rn(s, p); sa += fabsf(p); }
    out[t] = s; out_abs[t] = sa;
}

// RESCORE: one lane per listed candidate (the band), sequential arithmetic, weights read through a COLUMN-major copy (the CSC copy the
// library keeps for the selected-outputs route): a lane walks 3 KB contiguous, its neighbours other columns
__global__ void __launch_bounds__(256) rescore_pairs(const float* __restrict__ X, const float* __restrict__ Wcol, const float* __restrict__ bias,
                                                     const uint32_t* __restrict__ pair_q, const uint32_t* __restrict__ pair_col, uint64_t n_pairs, float* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_pairs) return;
    const float* __restrict__ x = X + (uint64_t)pair_q[t] * D;
    const float* __restrict__ w = Wcol + (uint64_t)pair_col[t] * D;
    float s = bias[pair_col[t]];
    for (int k = 0; k < D; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k), wv = *reinterpret_cast<const float4*>(w + k);
        s = __fadd_rn(s, __fmul_rn(xv.x, wv.x)); s = __fadd_rn(s, __fmul_rn(xv.y, wv.y)); s = __fadd_rn(s, __fmul_rn(xv.z, wv.z)); s = __fadd_rn(s, __fmul_rn(xv.w, wv.w));
    }
    out[t] = s;
}

int main(int argc, char** argv) {
    const uint32_t NQ = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 65536u, P = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1024u;
    const uint64_t ld = (uint64_t)P * C, n_items = (uint64_t)NQ * BEAM;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> X((size_t)NQ * D), W((size_t)D * ld), bias(ld);
    for (uint32_t q = 0; q < NQ; ++q) { double ss = 0; for (int k = 0; k < D; ++k) { X[(size_t)q * D + k] = nd(rng); ss += (double)X[(size_t)q * D + k] * X[(size_t)q * D + k]; }
        const float inv = (float)(1.0 / std::sqrt(ss)); for (int k = 0; k < D; ++k) X[(size_t)q * D + k] *= inv; }
    for (auto& v : W) v = nd(rng);
    for (auto& v : bias) v = nd(rng);
    // items: query q visits parents (q * 7 + j * 97) % P, tile-sorted, cut into blocks of <= 64 items of one tile
    std::vector<std::vector<uint32_t>> per_tile(P);
    for (uint32_t q = 0; q < NQ; ++q) for (int j = 0; j < BEAM; ++j) per_tile[(q * 7u + (uint32_t)j * 97u) % P].push_back(q);
    std::vector<Item> items; std::vector<uint32_t> b0, bn;
    for (uint32_t t = 0; t < P; ++t) for (size_t i = 0; i < per_tile[t].size(); ++i) { if (i % QB == 0) { b0.push_back((uint32_t)items.size()); bn.push_back((uint32_t)std::min<size_t>(QB, per_tile[t].size() - i)); } items.push_back(Item{per_tile[t][i], t}); }
    std::printf("shape: NQ=%u D=%d parents=%u cols/parent=%d items=%zu blocks=%zu  flops/pre-score=%.1f G\n", NQ, D, P, C, items.size(), b0.size(), 2.0 * items.size() * C * D / 1e9);
    float *dX, *dW, *dB, *dS, *dE, *dA; Item* dI; uint32_t *d0, *dn;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dB, bias.size() * 4)); CK(hipMalloc(&dI, items.size() * sizeof(Item)));
    CK(hipMalloc(&d0, b0.size() * 4)); CK(hipMalloc(&dn, bn.size() * 4)); CK(hipMalloc(&dS, n_items * C * 4)); CK(hipMalloc(&dE, n_items * C * 4)); CK(hipMalloc(&dA, n_items * C * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dI, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice)); CK(hipMemcpy(d0, b0.data(), b0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dn, bn.data(), bn.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& fn, int reps) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    const float ms_pre = timeit([&] { hipLaunchKernelGGL(prescore_mfma, dim3((uint32_t)b0.size()), dim3(256), 0, 0, dX, dW, ld, dB, dI, d0, dn, dS); }, 10);
    std::printf("pre-score (v_mfma_f32_16x16x4_f32): %.3f ms = %.1f TFLOP/s\n", ms_pre, 2.0 * items.size() * C * D / ms_pre / 1e9);
    const float ms_ex = timeit([&] { hipLaunchKernelGGL(exact_all, dim3((uint32_t)((n_items * C + 255) / 256)), dim3(256), 0, 0, dX, dW, ld, dB, dI, n_items, dE, dA); }, 2);
    std::printf("exact, every candidate, naive one-thread-per-candidate kernel (not K1G): %.3f ms = %.1f TFLOP/s\n", ms_ex, 2.0 * items.size() * C * D / ms_ex / 1e9);
    std::vector<float> S(n_items * C), E(n_items * C), A(n_items * C);
    CK(hipMemcpy(S.data(), dS, S.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(E.data(), dE, E.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(A.data(), dA, A.size() * 4, hipMemcpyDeviceToHost));
    const double u = std::ldexp(1.0, -24), cb = (3.0 * D + 2.0) * u;
    double worst = 0, mean_rel = 0; uint64_t viol = 0;
    for (size_t i = 0; i < S.size(); ++i) { const double d = std::fabs((double)S[i] - E[i]); const double r = d / (u * A[i]); worst = std::max(worst, r); mean_rel += r; if (d > cb * A[i]) ++viol; }
    std::printf("|s' - s| / (u * sum|xw|): max %.3f  mean %.4f   (proven bound: %.0f; violations of the bound: %llu of %zu)\n", worst, mean_rel / S.size(), 3.0 * D + 2.0, (unsigned long long)viol, S.size());
    // per query: its BEAM items' candidates, top-10 by exact vs the band procedure
    std::vector<double> wn(ld, 0.0);
    for (int k = 0; k < D; ++k) for (uint64_t c = 0; c < ld; ++c) wn[c] += (double)W[(size_t)k * ld + c] * W[(size_t)k * ld + c];
    for (auto& v : wn) v = std::sqrt(v);
    std::vector<std::vector<uint32_t>> q_items(NQ);
    for (uint32_t i = 0; i < items.size(); ++i) q_items[items[i].q].push_back(i);
    const int K = 10;
    uint64_t band_rig = 0, band_cs = 0, cands = 0, wrong = 0;
    std::vector<uint32_t> pq, pc;
    for (uint32_t q = 0; q < NQ; ++q) {
        std::vector<float> lo_r, lo_c; std::vector<std::pair<float, uint64_t>> ex;
        for (uint32_t it : q_items[q]) for (int c = 0; c < C; ++c) {
            const uint64_t t = (uint64_t)it * C + c; const double bc = cb * (std::fabs((double)bias[(uint64_t)items[it].tile * C + c]) + wn[(uint64_t)items[it].tile * C + c]);   // ||x|| = 1
            lo_r.push_back((float)(S[t] - cb * A[t])); lo_c.push_back((float)(S[t] - bc)); ex.emplace_back(E[t], t);
        }
        std::nth_element(lo_r.begin(), lo_r.begin() + K - 1, lo_r.end(), std::greater<float>()); const float thr_r = lo_r[K - 1];
        std::nth_element(lo_c.begin(), lo_c.begin() + K - 1, lo_c.end(), std::greater<float>()); const float thr_c = lo_c[K - 1];
        std::partial_sort(ex.begin(), ex.begin() + K, ex.end(), [](auto& a, auto& b) { return a.first > b.first; });
        for (uint32_t it : q_items[q]) for (int c = 0; c < C; ++c) {
            const uint64_t t = (uint64_t)it * C + c; const double bc = cb * (std::fabs((double)bias[(uint64_t)items[it].tile * C + c]) + wn[(uint64_t)items[it].tile * C + c]);
            ++cands;
            if (S[t] + cb * A[t] >= thr_r) ++band_rig;
            if (S[t] + bc >= thr_c) { ++band_cs; pq.push_back(q); pc.push_back((uint32_t)((uint64_t)items[it].tile * C + c)); }
        }
        for (int i = 0; i < K; ++i) { const uint64_t t = ex[i].second; const uint32_t it = (uint32_t)(t / C); const int c = (int)(t % C);
            const double bc = cb * (std::fabs((double)bias[(uint64_t)items[it].tile * C + c]) + wn[(uint64_t)items[it].tile * C + c]);
            if (!(S[t] + bc >= thr_c)) ++wrong; }                                   // an exact top-10 member outside the band would be a bug in the bound
    }
    std::printf("top-%d of %d candidates per query: band (must be re-scored exactly) = %.3f %% of the candidates with b = (3D+2)u*sum|xw| (%.2f per query), "
                "%.3f %% with the Cauchy-Schwarz bound (%.2f per query); exact top-%d members outside the band: %llu\n", K, C * BEAM, 100.0 * band_rig / cands,
                (double)band_rig / NQ, 100.0 * band_cs / cands, (double)band_cs / NQ, K, (unsigned long long)wrong);
    // rescore cost: column-major copy of W, one lane per band candidate
    std::vector<float> Wc((size_t)ld * D);
    for (int k = 0; k < D; ++k) for (uint64_t c = 0; c < ld; ++c) Wc[(size_t)c * D + k] = W[(size_t)k * ld + c];
    float* dWc; uint32_t *dpq, *dpc; float* dR;
    CK(hipMalloc(&dWc, Wc.size() * 4)); CK(hipMemcpy(dWc, Wc.data(), Wc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dpq, pq.size() * 4)); CK(hipMalloc(&dpc, pc.size() * 4)); CK(hipMalloc(&dR, pq.size() * 4));
    CK(hipMemcpy(dpq, pq.data(), pq.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpc, pc.data(), pc.size() * 4, hipMemcpyHostToDevice));
    const float ms_rs = timeit([&] { hipLaunchKernelGGL(rescore_pairs, dim3((uint32_t)((pq.size() + 255) / 256)), dim3(256), 0, 0, dX, dWc, dB, dpq, dpc, (uint64_t)pq.size(), dR); }, 5);
    std::vector<float> R(pq.size()); CK(hipMemcpy(R.data(), dR, R.size() * 4, hipMemcpyDeviceToHost));
    std::printf("re-score of the %zu band candidates (Cauchy-Schwarz band), one lane each, column-major weights: %.3f ms\n", pq.size(), ms_rs);
    std::printf("pre-score + re-score = %.3f ms for %u queries x %d parents; K1G (the exact VALU SGEMM) runs this many cells at 25-35 TFLOP/s = %.2f-%.2f ms\n", ms_pre + ms_rs, NQ, BEAM,
                2.0 * items.size() * C * D / 35e9, 2.0 * items.size() * C * D / 25e9);
    return 0;
}
