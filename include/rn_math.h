/* rn_math.h — deterministic fp32 transcendental functions shared by the HIP kernels
 * (hipcc, gfx950) and the C oracle (gcc, x86-64).
 *
 * Every operation below is an IEEE-754 single-precision add/mul/fma/rint/ldexp, issued in a
 * fixed order with explicit fmaf, so a gcc build (-ffp-contract=off) and a hipcc build
 * (-ffp-contract=off) return bit-identical results.  That is what lets the post-processing
 * chain (sigmoid -> per-class top-k -> NMS keep mask) be compared bit-for-bit between the
 * GPU path and the oracle instead of "within an ulp, unless a near-tie flips".
 *
 * The reference gets these functions from TensorFlow/Eigen (tf.nn.sigmoid, tf.math.exp,
 * tf.math.log: retinanet/model/layers/postprocessing_ops.py:99,114;
 * retinanet/dataloader/label_encoder.py:64; retinanet/losses/loss_impl.py:19,23), whose
 * last-bit behaviour is not reproducible without TF; tests/test_oracle_math.py pins these
 * against libm to <= 2 ulp (sigmoid, a composition, <= 3).
 */
#ifndef RN_MATH_H_
#define RN_MATH_H_

#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RN_HD __host__ __device__ __forceinline__
#else
#define RN_HD static inline
#endif

/* e^x, Cephes-style: n = rint(x*log2e), r = x - n*ln2 (two-step), degree-5 polynomial. */
RN_HD float rn_expf(float x) {
  if (x != x) return x;
  if (x > 88.7228394f) return INFINITY;
  if (x < -103.972084f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  p = fmaf(p, r2, r);
  p = p + 1.0f;
  return ldexpf(p, (int)n);
}

/* natural log, Cephes logf.  x <= 0 -> -inf / nan like libm. */
RN_HD float rn_logf(float x) {
  if (x != x) return x;
  if (x < 0.0f) return NAN;
  if (x == 0.0f) return -INFINITY;
  if (x == INFINITY) return x;
  int e;
  float m = frexpf(x, &e); /* m in [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  const float z = m * m;
  float y = 7.0376836292e-2f;
  y = fmaf(y, m, -1.1514610310e-1f);
  y = fmaf(y, m, 1.1676998740e-1f);
  y = fmaf(y, m, -1.2420140846e-1f);
  y = fmaf(y, m, 1.4249322787e-1f);
  y = fmaf(y, m, -1.6668057665e-1f);
  y = fmaf(y, m, 2.0000714765e-1f);
  y = fmaf(y, m, -2.4999993993e-1f);
  y = fmaf(y, m, 3.3333331174e-1f);
  y = y * m * z;
  const float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(-0.5f, z, y);
  float r = m + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

/* log(1+t) for t >= 0, accurate for tiny t (the u-1 correction trick). */
RN_HD float rn_log1pf_pos(float t) {
  const float u = 1.0f + t;
  if (u == 1.0f) return t;
  return rn_logf(u) * (t / (u - 1.0f));
}

/* logistic sigmoid 1/(1+e^-x) with one correctly rounded division. */
RN_HD float rn_sigmoidf(float x) {
  return 1.0f / (1.0f + rn_expf(-x));
}

#endif /* RN_MATH_H_ */
