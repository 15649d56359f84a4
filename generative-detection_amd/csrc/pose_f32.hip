// Pose-head loss terms of PoseLoss.forward in ONE launch: translation / yaw terms
// (src/modules/losses/contperceptual.py:111-132: L1 or L2 on t1..t3, smooth-L1 on sin(yaw)), sigmoid focal class loss (:176-181,
// mmdet FocalLoss defaults restated: gamma 2, alpha 0.25, mean over B*classes), box-size and fill-factor MSE (:183-191,205-212)
// and the per-class KL of the box posterior against the dataset-statistics prior (:193-203, with the reference's [8,1] x [1,8]
// broadcast: entry i sums the cross term over ALL prior dimensions j).  Every term is a masked mean over the samples whose class
// is not BACKGROUND_CLASS_IDX (:17,228).  The reference evaluates them with ~60 tiny torch launches and a host loop over the batch;
// here one block walks the batch, reduces, and also writes the Jacobians, so the backward is one more small launch.
#include "common.h"

namespace {

struct PoseP {
  const float* dec_pose;    // [B][8 + NC]: pose 4 | lhw 3 | fill 1 | class logits NC
  const float* pose_gt;     // [B][4]
  const float* bbox_gt;     // [B][3]
  const float* fill_gt;     // [B]
  const int64_t* class_gt;  // [B]
  const float* moments;     // [B][16]: box posterior mean 8 | raw logvar 8 (clamped to [-30, 20] as DiagonalGaussianDistribution does)
  const float* prior;       // [L][3][8]: mean | var | logvar per label
  const int* prior_idx;     // [B]: prior row of the sample's label, < 0 = label "background" (left out of the KL)
  int B, NC, L, bg_idx, l2, yaw;
  float gamma, alpha;
  float* out;               // [9]: pose, class, bbox, fill, kl | means of t1, t2, t3, v3
  float* jac_pose;          // [4][B][8 + NC]
  float* jac_mom;           // [B][16]
};

__device__ float block_sum(float v, float* red) {   // 256 threads; result broadcast to all
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float elem_loss(float d, int l2, float& grad) {
  if (l2) { grad = 2.f * d; return d * d; }
  grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return fabsf(d);
}

__global__ __launch_bounds__(256) void pose_losses_kernel(PoseP p) {
  __shared__ float red[4];
  const int W = 8 + p.NC;
  float cnt = 0.f;
  for (int b = threadIdx.x; b < p.B; b += 256) cnt += p.class_gt[b] != p.bg_idx ? 1.f : 0.f;
  const float nb = block_sum(cnt, red);
  const float inv_nb = nb > 0.f ? 1.f / nb : 0.f;       // sum / clamp(nb, 1) * (nb > 0)
  const float inv_cls = 1.f / ((float)p.B * (float)p.NC), inv_b = 1.f / (float)p.B;
  float s_pose = 0.f, s_cls = 0.f, s_box = 0.f, s_fill = 0.f, s_kl = 0.f, s_t[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = threadIdx.x; b < p.B; b += 256) {
    const float* dp = p.dec_pose + (int64_t)b * W;
    const int64_t cls = p.class_gt[b];
    const float mask = cls != p.bg_idx ? 1.f : 0.f;
    float* j0 = p.jac_pose + ((int64_t)0 * p.B + b) * W;
    float* j1 = p.jac_pose + ((int64_t)1 * p.B + b) * W;
    float* j2 = p.jac_pose + ((int64_t)2 * p.B + b) * W;
    float* j3 = p.jac_pose + ((int64_t)3 * p.B + b) * W;
    for (int c = 0; c < W; ++c) { j0[c] = 0.f; j1[c] = 0.f; j2[c] = 0.f; j3[c] = 0.f; }
    // translation + rotation (QUIRK :269: the call passes (gt, pred) into the (pred, gt) slots; every term is symmetric)
    float tsum = 0.f;
    for (int i = 0; i < 4; ++i) {
      float g, t;
      if (i == 3 && p.yaw) {
        const float d = __sinf(dp[3]) - __sinf(p.pose_gt[b * 4 + 3]);
        const float ad = fabsf(d);
        t = ad < 1.f ? 0.5f * d * d : ad - 0.5f;                               // SmoothL1Loss(beta = 1)
        g = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * __cosf(dp[3]);
      } else {
        t = elem_loss(dp[i] - p.pose_gt[b * 4 + i], p.l2, g);
      }
      s_t[i] += t;
      tsum += t;
      j0[i] = g * mask * inv_nb;
    }
    s_pose += tsum * mask;
    // sigmoid focal loss over the class logits; one_hot(target, NC + 1)[:, :NC]: class id NC has no positive column
    for (int k = 0; k < p.NC; ++k) {
      const float x = dp[8 + k], t = cls == k ? 1.f : 0.f;
      const float pr = 1.f / (1.f + __expf(-x));
      const float pt = (1.f - pr) * t + pr * (1.f - t);
      const float aw = p.alpha * t + (1.f - p.alpha) * (1.f - t);
      const float ptg1 = p.gamma == 2.f ? pt : __powf(pt, p.gamma - 1.f);
      const float fw = aw * ptg1 * pt;
      const float bce = fmaxf(x, 0.f) - x * t + __logf(1.f + __expf(-fabsf(x)));
      s_cls += bce * fw;
      j1[8 + k] = ((pr - t) * fw + bce * aw * p.gamma * ptg1 * (1.f - 2.f * t) * pr * (1.f - pr)) * inv_cls;
    }
    for (int j = 0; j < 3; ++j) {
      const float d = dp[4 + j] - p.bbox_gt[b * 3 + j];
      s_box += d * d * mask;
      j2[4 + j] = 2.f * d * mask * inv_nb;
    }
    {
      const float d = dp[7] - p.fill_gt[b];
      s_fill += d * d * mask;
      j3[7] = 2.f * d * mask * inv_nb;
    }
    // KL(posterior_i || prior_j) summed over j for every i (the reference's broadcast), then over i
    const int row = p.prior_idx[b];
    const float keep = row >= 0 ? 1.f : 0.f;
    const float* pm = p.prior + (int64_t)(row >= 0 ? row : 0) * 24;
    float inv_var_sum = 0.f, ol_sum = 0.f;
    for (int j = 0; j < 8; ++j) { inv_var_sum += 1.f / (pm[8 + j] + 1e-5f); ol_sum += pm[16 + j]; }
    for (int i = 0; i < 8; ++i) {
      const float m = p.moments[b * 16 + i], lv_raw = p.moments[b * 16 + 8 + i];
      const float lv = fminf(fmaxf(lv_raw, -30.f), 20.f);
      const float in_range = (lv_raw >= -30.f && lv_raw <= 20.f) ? 1.f : 0.f;
      const float ev = __expf(lv);
      float q = 0.f, dq = 0.f;
      for (int j = 0; j < 8; ++j) {
        const float d = m - pm[j], iv = 1.f / (pm[8 + j] + 1e-5f);
        q += d * d * iv;
        dq += d * iv;
      }
      s_kl += keep * 0.5f * (q + ev * inv_var_sum - 8.f - 8.f * lv + ol_sum);
      p.jac_mom[b * 16 + i] = keep * inv_nb * dq;
      p.jac_mom[b * 16 + 8 + i] = keep * inv_nb * 0.5f * (ev * inv_var_sum - 8.f) * in_range;
    }
  }
  const float r_pose = block_sum(s_pose, red), r_cls = block_sum(s_cls, red), r_box = block_sum(s_box, red);
  const float r_fill = block_sum(s_fill, red), r_kl = block_sum(s_kl, red);
  const float r_t0 = block_sum(s_t[0], red), r_t1 = block_sum(s_t[1], red), r_t2 = block_sum(s_t[2], red), r_t3 = block_sum(s_t[3], red);
  if (threadIdx.x == 0) {
    p.out[0] = r_pose * inv_nb; p.out[1] = r_cls * inv_cls; p.out[2] = r_box * inv_nb; p.out[3] = r_fill * inv_nb; p.out[4] = r_kl * inv_nb;
    p.out[5] = r_t0 * inv_b; p.out[6] = r_t1 * inv_b; p.out[7] = r_t2 * inv_b; p.out[8] = r_t3 * inv_b;
  }
}

// d_dec_pose = sum_k g[k] * jac_pose[k], d_moments = g[4] * jac_mom  (g = upstream gradient of the nine outputs)
__global__ void pose_losses_bwd_kernel(const float* __restrict__ g, const float* __restrict__ jac_pose, const float* __restrict__ jac_mom,
                                       int B, int W, float* __restrict__ d_dec_pose, float* __restrict__ d_moments) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = B * W;
  if (i < n) d_dec_pose[i] = g[0] * jac_pose[i] + g[1] * jac_pose[n + i] + g[2] * jac_pose[2 * n + i] + g[3] * jac_pose[3 * n + i];
  if (i < B * 16) d_moments[i] = g[4] * jac_mom[i];
}

}  // namespace

extern "C" {

int odvae_pose_losses_f32(const float* dec_pose, const float* pose_gt, const float* bbox_gt, const float* fill_gt, const int64_t* class_gt,
                          const float* moments, const float* prior, const int* prior_idx, int B, int NC, int L, int background_class_idx,
                          int pose_loss_l2, int train_on_yaw, float gamma, float alpha, float* out, float* jac_pose, float* jac_mom,
                          void* stream) {
  ODVAE_CHECK_ARG(dec_pose && pose_gt && bbox_gt && fill_gt && class_gt && moments && prior && prior_idx && out && jac_pose && jac_mom,
                  "pose_losses: null operand");
  ODVAE_CHECK_ARG(B > 0 && NC > 0 && L > 0, "pose_losses: empty shape B=%d NC=%d L=%d", B, NC, L);
  PoseP p{dec_pose, pose_gt, bbox_gt, fill_gt, class_gt, moments, prior, prior_idx, B, NC, L, background_class_idx, pose_loss_l2,
          train_on_yaw, gamma, alpha, out, jac_pose, jac_mom};
  hipLaunchKernelGGL(pose_losses_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  ODVAE_LAUNCH_CHECK("pose_losses");
  return ODVAE_OK;
}

int odvae_pose_losses_bwd_f32(const float* g, const float* jac_pose, const float* jac_mom, int B, int NC, float* d_dec_pose,
                              float* d_moments, void* stream) {
  ODVAE_CHECK_ARG(g && jac_pose && jac_mom && d_dec_pose && d_moments && B > 0 && NC > 0, "pose_losses_bwd: bad arguments");
  const int n = B * std::max(8 + NC, 16);
  hipLaunchKernelGGL(pose_losses_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g, jac_pose, jac_mom, B,
                     8 + NC, d_dec_pose, d_moments);
  ODVAE_LAUNCH_CHECK("pose_losses_bwd");
  return ODVAE_OK;
}

}  // extern "C"
