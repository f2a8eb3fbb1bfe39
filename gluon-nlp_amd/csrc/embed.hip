// Embedding front-end of the encoder stacks: word + token-type + learned-position gathers, add,
// LayerNorm -- one pass.  Reference: BertModel.get_initial_embedding (models/bert.py:458-504),
// ElectraModel.get_initial_embedding (models/electra.py:466-510), RobertaModel.get_initial_embedding
// (models/roberta.py:337-372); nn.Embedding and LearnedPositionalEmbedding (layers.py:460-482,
// np.take(..., mode='clip')) are row gathers.
//
// HBM-bound: one wave64 owns one token, the three table rows are read with 16-byte loads, summed in
// the reference's order ((word + type) + position, each sum rounded to the run dtype as the
// unfused ops would) and the row stays in registers between the statistics and the normalisation
// pass; fp32 two-pass statistics as in norm.hip.  Ids out of range are clipped (mode='clip').
#include "common.h"
#include "kernels.h"

namespace gnlp {

template <typename T, int MAXV>
__global__ void __launch_bounds__(256)
embed_norm_kernel(T *__restrict__ out, const void *__restrict__ ids, const void *__restrict__ types,
                  int id_dtype, const T *__restrict__ word_w, const T *__restrict__ type_w,
                  const T *__restrict__ pos_w, const T *__restrict__ gamma,
                  const T *__restrict__ beta, float eps, int64_t rows, int C, int64_t B, int64_t L,
                  int time_major, int64_t vocab, int64_t n_types, int64_t max_len) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C / N;
  const int64_t t = time_major ? row / B : row % L;
  int64_t id = load_len(ids, id_dtype, row);
  id = id < 0 ? 0 : (id > vocab - 1 ? vocab - 1 : id);
  const T *wr = word_w + id * (int64_t)C;
  const T *tr = nullptr, *pr = nullptr;
  if (type_w) {
    int64_t ty = types ? load_len(types, id_dtype, row) : 0;
    ty = ty < 0 ? 0 : (ty > n_types - 1 ? n_types - 1 : ty);
    tr = type_w + ty * (int64_t)C;
  }
  if (pos_w) pr = pos_w + (t > max_len - 1 ? max_len - 1 : t) * (int64_t)C;

  float vals[MAXV][N];
  float sum = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      V a = *reinterpret_cast<const V *>(wr + v * N);
#pragma unroll
      for (int q = 0; q < N; ++q) vals[i][q] = to_f32(a[q]);
      if (tr) {
        V b = *reinterpret_cast<const V *>(tr + v * N);
#pragma unroll
        for (int q = 0; q < N; ++q) vals[i][q] = to_f32(from_f32<T>(vals[i][q] + to_f32(b[q])));
      }
      if (pr) {
        V p = *reinterpret_cast<const V *>(pr + v * N);
#pragma unroll
        for (int q = 0; q < N; ++q) vals[i][q] = to_f32(from_f32<T>(vals[i][q] + to_f32(p[q])));
      }
#pragma unroll
      for (int q = 0; q < N; ++q) sum += vals[i][q];
    } else {
#pragma unroll
      for (int q = 0; q < N; ++q) vals[i][q] = 0.0f;
    }
  }
  T *orow = out + row * (int64_t)C;
  if (!gamma) {   // no embedding LayerNorm (RoBERTa with encoder_normalize_before = False)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + i * 64;
      if (v < nvec) {
        V o;
#pragma unroll
        for (int q = 0; q < N; ++q) o[q] = from_f32<T>(vals[i][q]);
        *reinterpret_cast<V *>(orow + v * N) = o;
      }
    }
    return;
  }
  const float mean = wave_sum(sum) / (float)C;
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (lane + i * 64 < nvec) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const float d = vals[i][q] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lane + i * 64;
    if (v < nvec) {
      V g = *reinterpret_cast<const V *>(gamma + v * N);
      V o;
      if (beta) {
        V bt = *reinterpret_cast<const V *>(beta + v * N);
#pragma unroll
        for (int q = 0; q < N; ++q)
          o[q] = from_f32<T>((vals[i][q] - mean) * rstd * to_f32(g[q]) + to_f32(bt[q]));
      } else {
#pragma unroll
        for (int q = 0; q < N; ++q) o[q] = from_f32<T>((vals[i][q] - mean) * rstd * to_f32(g[q]));
      }
      *reinterpret_cast<V *>(orow + v * N) = o;
    }
  }
}

template <typename T>
static bool embed_t(void *out, const void *ids, const void *types, int id_dtype, const void *word_w,
                    const void *type_w, const void *pos_w, const void *gamma, const void *beta,
                    float eps, int64_t B, int64_t L, int64_t C, int time_major, int64_t vocab,
                    int64_t n_types, int64_t max_len, hipStream_t s) {
  constexpr int N = VecOf<T>::N;
  constexpr int MAXV = 32 / N;   // 32 fp32 values per lane => C <= 2048
  const int64_t rows = B * L;
  if (rows == 0 || C == 0) return true;
  uintptr_t al = (uintptr_t)out | (uintptr_t)word_w | (uintptr_t)(type_w ? type_w : word_w) |
                 (uintptr_t)(pos_w ? pos_w : word_w) | (uintptr_t)(gamma ? gamma : word_w) |
                 (uintptr_t)(beta ? beta : word_w);
  if (C % N != 0 || C > 64 * N * MAXV || al % 16 != 0) return false;
  dim3 grid((unsigned)((rows + 3) / 4));
  embed_norm_kernel<T, MAXV><<<grid, 256, 0, s>>>(
      (T *)out, ids, types, id_dtype, (const T *)word_w, (const T *)type_w, (const T *)pos_w,
      (const T *)gamma, (const T *)beta, eps, rows, (int)C, B, L, time_major, vocab, n_types,
      max_len);
  return true;
}

bool launch_embedding(void *out, const void *ids, const void *types, int id_dtype,
                      const void *word_w, const void *type_w, const void *pos_w, const void *gamma,
                      const void *beta, int dtype, float eps, int64_t B, int64_t L, int64_t C,
                      int time_major, int64_t vocab, int64_t n_types, int64_t max_len,
                      hipStream_t s) {
  if (dtype == GNLP_BF16)
    return embed_t<bf16>(out, ids, types, id_dtype, word_w, type_w, pos_w, gamma, beta, eps, B, L,
                         C, time_major, vocab, n_types, max_len, s);
  return embed_t<float>(out, ids, types, id_dtype, word_w, type_w, pos_w, gamma, beta, eps, B, L, C,
                        time_major, vocab, n_types, max_len, s);
}

}  // namespace gnlp
