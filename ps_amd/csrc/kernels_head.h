// kernels_head.h -- the head of one sample (wide LR + add + clipped sigmoid + cross-entropy term + delta_L), as a device
// function: shared by the head's own kernels (kernels_emb.hip: k_head, k_last_bwd<HEAD>) and by the first delta GEMM of a
// training step, which computes its rows' delta_L itself instead of waiting for the head's launch (kernels_gemm.hip, round 4).
#pragma once
#include "kernels_emb.h"
#ifndef HEAD_T
#define HEAD_T(k) do { } while (0)
#endif
// ---------------------------------------------------------------------------
// head: wide LR + add + clipped sigmoid + cross-entropy term + delta_L
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_clip_d(float x) {
    return (float)(0.001f + (double)(.999f - 0.001f) / (1.0 + exp(-(double)x)));
}


// Eight lanes per sample, eight samples per wave: every load of the head is independent of the others (the out = 1
// layer's dot product: 34 strided loads per lane at K = 257; the wide part: id -> weight for 26 fields, four per
// lane) and the only serial pieces are two 3-step butterflies and the reference's sequential f32 sum of the F wide
// weights (layer/LRLayer.java:73-84), done with shuffles inside the group.  (One WAVE per sample was as fast per
// launch, but eight samples per wave is what lets a workgroup do the head of all the rows whose backward it owns.)
// valid = false: the lanes take part in the shuffles with sample b clamped, and store nothing.
// Returns delta_L * sigmoid' of the sample (every lane of the group holds it); 0 when there are no labels.
__device__ __forceinline__ float head_one(const HeadArgs &a, int b, int lane, bool valid) {
    const int l8 = lane & 7, gbase = lane & ~7;
    // Load order, all branch-free so that the compiler's in-order s_waitcnt bookkeeping stays exact:
    //   wide ids (F <= 32: the usual case; more fields fall back to the loop below) -> the row of the last layer's
    //   input and its weights -> (ids arrived) the wide weights -> dot product -> wide sum.
    // Every round trip overlaps the next one; the stores (touched marks, error count) wait until the end.
    // (With the id -> weight chain issued as one block BEFORE the row loads the head was faster on cache-resident
    // batches and slower on fresh ones: the row loads sat behind the wait for the ids.)
    int64_t wid[4] = {0, 0, 0, 0};
    float ww[4] = {0.f, 0.f, 0.f, 0.f};
    bool wbad = false;
    const bool wide_early = a.wide && a.F <= 32;
    if (wide_early) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 8 * r + l8;
            wid[r] = a.wide_ids[(size_t)b * a.F + (f < a.F ? f : a.F - 1)];
        }
    }
    float zl;                                               // the last FcLayer's activation for this sample
    if (a.a_last) {
        // FcLayer.forward with out = 1 (layer/FcLayer.java:76-77): the group's 8 lanes read 128 contiguous bytes
        // of the row per load, 8 loads per lane in flight (K <= 256: ONE memory round trip; scalar loads strided
        // over the lanes, 34 per lane behind a runtime trip count, took 7.6 of the head's 11.5 us), butterfly sum
        const float *__restrict__ x = a.a_last + (size_t)b * a.lda_last;
        const float *__restrict__ wl = a.w_last;
        const int k4 = a.k_last & ~3;
        float acc = 0.f;
        for (int kb = 0; kb < k4; kb += 256) {
            float4 xv[8], wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = kb + 32 * i + 4 * l8;
                const int kk = k < k4 ? k : 0;
                xv[i] = *reinterpret_cast<const float4 *>(x + kk);
                wv[i] = *reinterpret_cast<const float4 *>(wl + kk);
            }
            if (kb == 0 && wide_early) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool bad = wid[r] < 0 || wid[r] >= a.wide_rows;
                    wbad |= bad && 8 * r + l8 < a.F;
                    if (bad) wid[r] = 0;
                    ww[r] = a.wide_w[wid[r]];
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (kb + 32 * i + 4 * l8 >= k4) xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                acc += xv[i].x * wv[i].x; acc += xv[i].y * wv[i].y; acc += xv[i].z * wv[i].z; acc += xv[i].w * wv[i].w;
            }
        }
        for (int k = k4 + l8; k < a.k_last; k += 8) acc += x[k] * wl[k];       // <= 3 elements (the ones column)
#pragma unroll
        for (int off = 4; off; off >>= 1) acc += __shfl_xor(acc, off);
        zl = a.last_sigmoid ? sigmoid_clip_d(acc) : acc;
        if (valid && l8 == 0) a.zout[(size_t)b * a.ldz] = zl;
    } else {
        zl = a.zlast[(size_t)b * a.ldz];
        if (wide_early) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool bad = wid[r] < 0 || wid[r] >= a.wide_rows;
                wbad |= bad && 8 * r + l8 < a.F;
                if (bad) wid[r] = 0;
                ww[r] = a.wide_w[wid[r]];
            }
        }
    }
    if (wide_early) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (8 * r + l8 >= a.F) ww[r] = 0.f;
    }
    HEAD_T(4);
    float p;
    if (a.wide) {
        // LRLayer.forward (layer/LRLayer.java:73-84): sum over the F wide ids, sequential, then + bias
        float sumW = 0.f;
        for (int j0 = 0; j0 < a.F; j0 += 32) {
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w[r] = ww[r];
                const int f = j0 + 8 * r + l8;
                if (!wide_early && f < a.F) {
                    int64_t id = a.wide_ids[(size_t)b * a.F + f];
                    if (id < 0 || id >= a.wide_rows) { if (valid) atomicAdd(a.err, 1); id = 0; }
                    w[r] = a.wide_w[id];
                    if (valid && a.touched && a.train) a.touched[id] = 1;   // LRLayer.weights.put (never cleared)
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = a.F - j0 - 8 * r;
#pragma unroll
                for (int j = 0; j < 8; ++j) {                                    // field order; the 8 shuffles are independent
                    const float v = __shfl(w[r], gbase + j);
                    if (j < n) sumW += v;
                }
            }
        }
        sumW += a.wide_bias[0];
        if (wide_early && valid) {
            if (wbad) atomicAdd(a.err, 1);
            if (a.touched && a.train) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (8 * r + l8 < a.F) a.touched[wid[r]] = 1;   // LRLayer.weights.put (never cleared)
            }
        }
        HEAD_T(5);
        if (valid && l8 == 0) a.wide_z[b] = sumW;
        const float z = zl + sumW;                          // AddLayer.forward l.add(r)
        p = sigmoid_clip_d(z);
    } else {
        p = zl;                                             // last FcLayer already applied the sigmoid
    }
    HEAD_T(6);
    if (valid && l8 == 0) a.P[b] = p;
    if (!a.labels) return 0.f;
    const float l = a.labels[b];
    float d = (p - l) / (p * (1 - p));                      // loss/CrossEntropy.java:25
    d *= p * (1 - p);                                       // Sigmoid.backward (activations/Sigmoid.java:18)
    if (valid && l8 == 0) {
        // loss/CrossEntropy.java:15 (FastMath.log ~ log; double math, cast to float)
        a.terms[b] = (float)(-l * log((double)p) - ((1 - l) * log((double)(1 - p))));
        a.dlast[(size_t)b * a.ldd] = d;
    }
    return d;
}

