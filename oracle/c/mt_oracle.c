/*
 * ORACLE — test infrastructure only (never linked into the product).
 * Plain-C restatement of the reference's per-layer algorithm on NCDHW fp32 tensors, independent of PyTorch:
 * straightforward loops, double accumulators.  Each function cites the reference code it restates
 * (paths relative to the MIC-DKFZ/MultiTalent checkout).  Checked against golden vectors generated from the real
 * reference in tests/test_oracle_golden.py::test_c_oracle_*.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* nn.Conv3d(bias) — generic_UNet.py:57,67 (ConvDropoutNormNonlin.conv), conv_blocks.py:166-170 (BasicResidualBlock) */
void mto_conv3d(const float* x, const float* w, const float* b, float* y, int N, int Ci, int Di, int Hi, int Wi, int Co,
                int KD, int KH, int KW, int SD, int SH, int SW, int PD, int PH, int PW) {
  const int Do = (Di + 2 * PD - KD) / SD + 1, Ho = (Hi + 2 * PH - KH) / SH + 1, Wo = (Wi + 2 * PW - KW) / SW + 1;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int od = 0; od < Do; ++od)
        for (int oh = 0; oh < Ho; ++oh)
          for (int ow = 0; ow < Wo; ++ow) {
            double acc = b ? b[co] : 0.0;
            for (int ci = 0; ci < Ci; ++ci)
              for (int kd = 0; kd < KD; ++kd) {
                const int id = od * SD + kd - PD;
                if (id < 0 || id >= Di) continue;
                for (int kh = 0; kh < KH; ++kh) {
                  const int ih = oh * SH + kh - PH;
                  if (ih < 0 || ih >= Hi) continue;
                  for (int kw = 0; kw < KW; ++kw) {
                    const int iw = ow * SW + kw - PW;
                    if (iw < 0 || iw >= Wi) continue;
                    acc += (double)x[(((size_t)(n * Ci + ci) * Di + id) * Hi + ih) * Wi + iw] *
                           (double)w[((((size_t)co * Ci + ci) * KD + kd) * KH + kh) * KW + kw];
                  }
                }
              }
            y[(((size_t)(n * Co + co) * Do + od) * Ho + oh) * Wo + ow] = (float)acc;
          }
}

/* nn.ConvTranspose3d(kernel == stride, bias=False) — generic_UNet.py:335-336; weight [Ci][Co][k...] */
void mto_tconv3d(const float* x, const float* w, float* y, int N, int Ci, int Di, int Hi, int Wi, int Co, int KD, int KH, int KW) {
  const int Do = Di * KD, Ho = Hi * KH, Wo = Wi * KW;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int od = 0; od < Do; ++od)
        for (int oh = 0; oh < Ho; ++oh)
          for (int ow = 0; ow < Wo; ++ow) {
            const int id = od / KD, kd = od % KD, ih = oh / KH, kh = oh % KH, iw = ow / KW, kw = ow % KW;
            double acc = 0.0;
            for (int ci = 0; ci < Ci; ++ci)
              acc += (double)x[(((size_t)(n * Ci + ci) * Di + id) * Hi + ih) * Wi + iw] *
                     (double)w[((((size_t)ci * Co + co) * KD + kd) * KH + kh) * KW + kw];
            y[(((size_t)(n * Co + co) * Do + od) * Ho + oh) * Wo + ow] = (float)acc;
          }
}

/* nn.InstanceNorm3d(eps, affine, biased variance, no running stats) + nn.LeakyReLU(slope) in place —
 * generic_UNet.py:63-64,69-70 */
void mto_instnorm_lrelu(float* x, const float* gamma, const float* beta, int N, int C, long V, float eps, float slope) {
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float* p = x + ((size_t)n * C + c) * V;
      double s = 0.0, ss = 0.0;
      for (long v = 0; v < V; ++v) { s += p[v]; }
      const double mean = s / (double)V;
      for (long v = 0; v < V; ++v) { const double d = p[v] - mean; ss += d * d; }
      const double rstd = 1.0 / sqrt(ss / (double)V + (double)eps);
      for (long v = 0; v < V; ++v) {
        const double z = (p[v] - mean) * rstd * (double)gamma[c] + (double)beta[c];
        p[v] = (float)(z > 0 ? z : z * (double)slope);
      }
    }
}

/* MultiTalent loss statistics of one level — MultiTalent_Trainer_DDP.py:574-594: for valid channel c of sample b:
 * stats[b][c] = (sum_v BCEWithLogits(x, y), sum sig*y, sum sig*(1-y), sum (1-sig)*y), y = target in labels(c) (lut bitmask). */
void mto_multitalent_stats(const float* logits, const float* target, int B, int C, long V, const uint64_t* valid,
                           const uint64_t* lut, double* stats) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      double bce = 0, tp = 0, fp = 0, fn = 0;
      if ((valid[b] >> c) & 1ull) {
        for (long v = 0; v < V; ++v) {
          const double x = logits[((size_t)b * C + c) * V + v];
          const int lab = (int)target[(size_t)b * V + v];
          const double y = (lab >= 0 && lab < 64 && ((lut[c] >> lab) & 1ull)) ? 1.0 : 0.0;
          bce += fmax(x, 0.0) - x * y + log1p(exp(-fabs(x)));
          const double s = 1.0 / (1.0 + exp(-x));
          tp += s * y; fp += s * (1 - y); fn += (1 - s) * y;
        }
      }
      double* o = stats + ((size_t)b * C + c) * 4;
      o[0] = bce; o[1] = tp; o[2] = fp; o[3] = fn;
    }
}

/* softmax Dice+CE statistics of one level — dice_loss.py:117-151, crossentropy.py:8-11:
 * stats[b][c] = (ce_sum in slot c==0, tp, fp, fn) */
void mto_softmax_stats(const float* logits, const float* target, int B, int C, long V, double* stats) {
  for (int b = 0; b < B; ++b) {
    for (int c = 0; c < C * 4; ++c) stats[(size_t)b * C * 4 + c] = 0.0;
    for (long v = 0; v < V; ++v) {
      double mx = -1e300, se = 0.0;
      for (int c = 0; c < C; ++c) { const double x = logits[((size_t)b * C + c) * V + v]; if (x > mx) mx = x; }
      for (int c = 0; c < C; ++c) se += exp(logits[((size_t)b * C + c) * V + v] - mx);
      const int lab = (int)target[(size_t)b * V + v];
      for (int c = 0; c < C; ++c) {
        const double p = exp(logits[((size_t)b * C + c) * V + v] - mx) / se;
        const double y = (lab == c) ? 1.0 : 0.0;
        double* o = stats + ((size_t)b * C + c) * 4;
        o[1] += p * y; o[2] += p * (1 - y); o[3] += (1 - p) * y;
        if (lab == c) stats[(size_t)b * C * 4] -= log(p);
      }
    }
  }
}

/* SGD(nesterov, dampening 0) after clip_grad_norm_(max_norm) — nnUNetTrainerV2.py:166-170,254 */
void mto_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float wd, float mom, int first, float max_norm) {
  double ss = 0.0;
  for (long i = 0; i < n; ++i) ss += (double)g[i] * g[i];
  double coef = max_norm / (sqrt(ss) + 1e-6);
  if (coef > 1.0) coef = 1.0;
  for (long i = 0; i < n; ++i) {
    double d = g[i] * coef + (double)wd * p[i];
    double b = first ? d : (double)mom * buf[i] + d;
    buf[i] = (float)b;
    p[i] = (float)(p[i] - (double)lr * (d + (double)mom * b));
  }
}
