/* CPU oracle (TEST INFRASTRUCTURE ONLY; never linked into the product library).
 *
 * Plain-C restatement of gluonnlp.op.relative_position_bucket
 * (/root/reference/src/gluonnlp/op.py:229-286) using glibc's single-precision logf, i.e. the
 * arithmetic MXNet's CPU `np.log` on a float32 array performs (mshadow_op::log -> ::logf).
 * Used to cross-check the numpy restatement in oracle/ref.py (numpy's float32 log is its own
 * SIMD routine) on every relative position the encoder configs can produce.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

void oracle_relative_position_bucket(const int32_t *rel, int32_t *out, int64_t n,
                                     int bidirectional, int num_buckets, int max_distance) {
  int nb = num_buckets;
  if (bidirectional) nb /= 2;                       /* op.py:266 */
  const int max_exact = nb / 2;                     /* op.py:276 */
  const float denom = (float)log((double)max_distance / (double)max_exact); /* math.log, :283 */
  for (int64_t i = 0; i < n; ++i) {
    int32_t rp = -rel[i];                           /* op.py:262 */
    int32_t ret = 0;
    if (bidirectional) {
      if (rp < 0) ret += nb;                        /* op.py:267 */
      rp = abs(rp);                                 /* op.py:268 */
    } else if (rp < 0) {
      rp = 0;                                       /* op.py:271 */
    }
    int32_t v;
    if (rp < max_exact) {
      v = rp;                                       /* op.py:277,285 */
    } else {
      float r = logf((float)rp / (float)max_exact) / denom * (float)(nb - max_exact);
      v = max_exact + (int32_t)r;                   /* truncation toward zero, :281-283 */
      if (v > nb - 1) v = nb - 1;                   /* op.py:284 */
    }
    out[i] = ret + v;
  }
}
