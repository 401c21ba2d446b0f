// Scoring head on K logits per query (gfx950): pointwise sigmoid/BCE, two-head softmax/CE, listwise
// softmax/CE (= "negative_sampling"), plus the descending, retrieval-order-stable rank.
//
// Reference anchors (/root/reference/): labels + loss src/models/rerank/utils.py:208-254 (BCEWithLogits
// with pos_weight | CrossEntropy(weight=[1,pos_weight]) over [l1,l2] | CrossEntropy(target 0) over
// logits.view(Bq,K)); rank = `sorted(zip(docs, logits), key=score, reverse=True)`
// src/executors/Reranker_base_executor.py:923-935 (stable: ties keep retrieval order).
#include "rr_common.h"
#include <math.h>

namespace {

constexpr int MAXK = 4096;   // candidates per query handled by one workgroup

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
  return t;
}

// "a ranks before b": higher logit first, equal logits keep ascending candidate index.
__device__ __forceinline__ bool before(float sa, int ia, float sb, int ib) {
  return sa > sb || (sa == sb && ia < ib);
}

// one workgroup per query
__global__ __launch_bounds__(256) void head_query_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ logits_first,
                                                         const float* __restrict__ labels, int K, int loss_kind,
                                                         float pos_weight, int has_pw, float* __restrict__ scores,
                                                         int32_t* __restrict__ order,
                                                         float* __restrict__ part_loss, float* __restrict__ part_w) {
  __shared__ float sk[MAXK];
  __shared__ int si[MAXK];
  __shared__ float red[4];
  const int qi = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (size_t)qi * K;
  const float* x1 = logits_first ? logits_first + (size_t)qi * K : nullptr;
  const float* y = labels ? labels + (size_t)qi * K : nullptr;

  float lsum = 0.f, wsum = 0.f;
  if (loss_kind == 0) {          // BCE with logits, optional pos_weight; mean over all N later
    for (int i = tid; i < K; i += blockDim.x) {
      const float xi = x[i], yi = y ? y[i] : (i == 0 ? 1.f : 0.f);
      const float lw = has_pw ? 1.f + (pos_weight - 1.f) * yi : 1.f;
      lsum += (1.f - yi) * xi + lw * (log1pf(expf(-fabsf(xi))) + fmaxf(-xi, 0.f));
      if (scores) scores[(size_t)qi * K + i] = 1.f / (1.f + expf(-xi));
    }
    wsum = (float)K;             // reduced as a count
  } else if (loss_kind == 1) {   // two heads: CE over [l1, l2], class weights [1, pos_weight]
    for (int i = tid; i < K; i += blockDim.x) {
      const float a = x1[i], b = x[i], yi = y ? y[i] : (i == 0 ? 1.f : 0.f);
      const float mx = fmaxf(a, b), lse = mx + logf(expf(a - mx) + expf(b - mx));
      const float w = (has_pw && yi != 0.f) ? pos_weight : 1.f;
      lsum += w * (lse - (yi != 0.f ? b : a));
      wsum += w;
      if (scores) scores[(size_t)qi * K + i] = 1.f / (1.f + expf(a - b));
    }
  } else if (loss_kind == 3) {   // RerankModel quirk: loss_fn(logits, logits) with 2 heads = CE with the logits
    // themselves as class-probability targets (rerank_model.py:328): -sum_c w_c t_c log_softmax(x)_c, mean over N
    for (int i = tid; i < K; i += blockDim.x) {
      const float a = x1[i], b = x[i];
      const float mx = fmaxf(a, b), lse = mx + logf(expf(a - mx) + expf(b - mx));
      const float w1 = has_pw ? pos_weight : 1.f;
      lsum += -(a * (a - lse) + w1 * b * (b - lse));
      if (scores) scores[(size_t)qi * K + i] = 1.f / (1.f + expf(a - b));
    }
    wsum = (float)K;
  } else {                       // listwise: CE(target = candidate 0) over the K logits
    float mx = -INFINITY;
    for (int i = tid; i < K; i += blockDim.x) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int i = tid; i < K; i += blockDim.x) se += expf(x[i] - mx);
    se = block_sum(se, red);
    if (scores)
      for (int i = tid; i < K; i += blockDim.x) scores[(size_t)qi * K + i] = expf(x[i] - mx) / se;
    if (tid == 0) lsum = mx + logf(se) - x[0];
    wsum = tid == 0 ? 1.f : 0.f;
  }
  if (part_loss) {
    const float L = block_sum(lsum, red);
    const float Wt = (loss_kind == 0 || loss_kind == 3) ? (float)K : block_sum(wsum, red);
    if (tid == 0) { part_loss[qi] = L; part_w[qi] = Wt; }
  }

  if (order) {                   // bitonic sort of (logit desc, index asc) on the next power of two
    int n2 = 1;
    while (n2 < K) n2 <<= 1;
    for (int i = tid; i < n2; i += blockDim.x) {
      sk[i] = i < K ? x[i] : -INFINITY;
      si[i] = i < K ? i : 0x7fffffff;   // padding ranks after everything, including real -inf logits
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < n2; i += blockDim.x) {
          const int l = i ^ j;
          if (l > i) {
            const bool up = (i & k) == 0;   // ascending-rank run
            const float sa = sk[i], sb = sk[l];
            const int ia = si[i], ib = si[l];
            const bool a_first = before(sa, ia, sb, ib);
            if (up ? !a_first : a_first) { sk[i] = sb; sk[l] = sa; si[i] = ib; si[l] = ia; }
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < K; i += blockDim.x) order[(size_t)qi * K + i] = si[i];
  }
}

// fixed-order reduction of the per-query partials -> scalar loss (bitwise reproducible)
__global__ void head_reduce_kernel(const float* __restrict__ part_loss, const float* __restrict__ part_w, int Bq,
                                   float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double L = 0.0, Wt = 0.0;
    for (int i = 0; i < Bq; ++i) { L += part_loss[i]; Wt += part_w[i]; }
    loss[0] = (float)(L / Wt);
  }
}

}  // namespace

hipError_t rr_launch_head(const float* logits, const float* logits_first, const float* labels, int Bq, int K,
                          int loss_kind, float pos_weight, int has_pw, float* scores, int32_t* order, float* loss,
                          float* part_loss, float* part_w, hipStream_t st) {
  if (Bq <= 0 || K <= 0 || K > MAXK) return hipErrorInvalidValue;
  if ((loss_kind == 1 || loss_kind == 3) && !logits_first) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_query_kernel, dim3(Bq), dim3(256), 0, st, logits, logits_first, labels, K, loss_kind,
                     pos_weight, has_pw, scores, order, loss ? part_loss : nullptr, part_w);
  if (loss) hipLaunchKernelGGL(head_reduce_kernel, dim3(1), dim3(64), 0, st, part_loss, part_w, Bq, loss);
  return hipGetLastError();
}
