// Waypoint head: 4 autoregressive GRUCell(2,64) steps + Linear(64,2) + L1 loss, forward and
// backward, one workgroup (one wave) per sample — the whole recurrence lives in registers/LDS
// (model_vec.py:666-680 and run_steps/phase2_train_net.py:104).  The reference issues ~30 tiny
// launches forward and as many backward for this; here it is one launch each way.
//
// GRUCell semantics: gates r,z,n in weight_ih[192,2] / weight_hh[192,64];
//   r = s(Wir x + bir + Whr h + bhr), z = s(Wiz x + biz + Whz h + bhz),
//   n = tanh(Win x + bin + r * (Whn h + bhn)),  h' = n + z * (h - n)
// Waypoint recurrence: x0 = 0; h = GRU(x + target, h); x = x + Wo h + bo; out = [x1..x4].
#include "common.h"

namespace {
constexpr int HID = 64;
constexpr int STEPS_MAX = 8;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// saved per sample: hs[(S+1)*64], gates[S*4*64] = {r,z,n,ghn}, xin[S*2]
__global__ __launch_bounds__(HID) void gru_head_fwd_kernel(const float* __restrict__ z0, const float* __restrict__ target,
                                                           const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                           const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                           const float* __restrict__ gt, float* __restrict__ pred,
                                                           float* __restrict__ hs, float* __restrict__ gates,
                                                           float* __restrict__ xin_save, float* __restrict__ loss_terms,
                                                           int steps) {
  const int b = blockIdx.x, j = threadIdx.x;
  __shared__ float sh[HID];
  __shared__ float sx[2];
  float h = z0[(size_t)b * HID + j];
  float x0 = 0.f, x1 = 0.f;
  const float t0 = target[b * 2], t1 = target[b * 2 + 1];
  if (hs) hs[((size_t)b * (steps + 1)) * HID + j] = h;
  float lsum = 0.f;
  for (int t = 0; t < steps; ++t) {
    const float xi0 = x0 + t0, xi1 = x1 + t1;
    sh[j] = h;
    __syncthreads();
    float gi[3], gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int row = g * HID + j;
      gi[g] = w_ih[row * 2] * xi0 + w_ih[row * 2 + 1] * xi1 + b_ih[row];
      float acc = 0.f;
      const float* wr = w_hh + (size_t)row * HID;
      for (int k = 0; k < HID; ++k) acc += wr[k] * sh[k];
      gh[g] = acc + b_hh[row];
    }
    const float r = sigmoidf_(gi[0] + gh[0]);
    const float z = sigmoidf_(gi[1] + gh[1]);
    const float n = tanhf(gi[2] + r * gh[2]);
    h = n + z * (h - n);
    if (gates) {
      float* gsv = gates + (((size_t)b * steps + t) * 4) * HID;
      gsv[j] = r; gsv[HID + j] = z; gsv[2 * HID + j] = n; gsv[3 * HID + j] = gh[2];
      hs[((size_t)b * (steps + 1) + t + 1) * HID + j] = h;
      if (j < 2) xin_save[((size_t)b * steps + t) * 2 + j] = j ? xi1 : xi0;
    }
    __syncthreads();
    sh[j] = h;
    __syncthreads();
    if (j < 2) {
      float acc = 0.f;
      for (int k = 0; k < HID; ++k) acc += w_out[j * HID + k] * sh[k];
      sx[j] = acc + b_out[j];
    }
    __syncthreads();
    x0 = sx[0] + x0;
    x1 = sx[1] + x1;
    if (j < 2) {
      const float xv = j ? x1 : x0;
      pred[((size_t)b * steps + t) * 2 + j] = xv;
      if (gt) lsum += fabsf(xv - gt[((size_t)b * steps + t) * 2 + j]);
    }
    __syncthreads();
  }
  if (loss_terms) {
    if (j < 2) sx[j] = lsum;
    __syncthreads();
    if (j == 0) loss_terms[b] = sx[0] + sx[1];
  }
}

__global__ void loss_finalize_kernel(const float* __restrict__ terms, int B, float inv_count, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0;
    for (int b = 0; b < B; ++b) s += terms[b];
    *loss = (float)(s * inv_count);
  }
}

// parameter-gradient partial layout per sample
constexpr int OFF_WIH = 0, OFF_WHH = 384, OFF_BIH = 384 + 12288, OFF_BHH = OFF_BIH + 192, OFF_WOUT = OFF_BHH + 192,
              OFF_BOUT = OFF_WOUT + 128, NPART = OFF_BOUT + 2;

__global__ __launch_bounds__(HID) void gru_head_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           const float* __restrict__ dpred_in, float gscale,
                                                           const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                           const float* __restrict__ w_out, const float* __restrict__ hs,
                                                           const float* __restrict__ gates, const float* __restrict__ xin_save,
                                                           float* __restrict__ dz0, float* __restrict__ part, int steps) {
  const int b = blockIdx.x, j = threadIdx.x;
  __shared__ float sd[3 * HID];  // gate pre-activation grads (r, z, hn-path) of the current step
  __shared__ float sred[2][HID];
  __shared__ float shp[HID];   // h_{t-1} of all units (for dW_hh)
  float* P = part + (size_t)b * NPART;
  for (int i = j; i < NPART; i += HID) P[i] = 0.f;
  __syncthreads();
  float gh = 0.f;              // grad wrt h_t (this thread's unit)
  float gx0 = 0.f, gx1 = 0.f;  // grad wrt x_t
  for (int t = steps - 1; t >= 0; --t) {
    // dL/dpred[t]
    float dp0, dp1;
    if (dpred_in) {
      dp0 = dpred_in[((size_t)b * steps + t) * 2] * gscale;
      dp1 = dpred_in[((size_t)b * steps + t) * 2 + 1] * gscale;
    } else {
      const float e0 = pred[((size_t)b * steps + t) * 2] - gt[((size_t)b * steps + t) * 2];
      const float e1 = pred[((size_t)b * steps + t) * 2 + 1] - gt[((size_t)b * steps + t) * 2 + 1];
      dp0 = (e0 > 0.f ? 1.f : (e0 < 0.f ? -1.f : 0.f)) * gscale;
      dp1 = (e1 > 0.f ? 1.f : (e1 < 0.f ? -1.f : 0.f)) * gscale;
    }
    gx0 += dp0;
    gx1 += dp1;
    const float ht = hs[((size_t)b * (steps + 1) + t + 1) * HID + j];
    const float hp = hs[((size_t)b * (steps + 1) + t) * HID + j];
    // x_t = x_{t-1} + W_o h_t + b_o
    P[OFF_WOUT + j] += gx0 * ht;
    P[OFF_WOUT + HID + j] += gx1 * ht;
    if (j < 2) P[OFF_BOUT + j] += j ? gx1 : gx0;
    gh += w_out[j] * gx0 + w_out[HID + j] * gx1;
    // GRU cell backward
    const float* gsv = gates + (((size_t)b * steps + t) * 4) * HID;
    const float r = gsv[j], z = gsv[HID + j], n = gsv[2 * HID + j], ghn = gsv[3 * HID + j];
    const float dn = gh * (1.f - z);
    const float dz = gh * (hp - n);
    float dh_prev = gh * z;
    const float da_n = dn * (1.f - n * n);
    const float dr = da_n * ghn;
    const float da_r = dr * r * (1.f - r);
    const float da_z = dz * z * (1.f - z);
    const float dghn = da_n * r;
    const float xi0 = xin_save[((size_t)b * steps + t) * 2], xi1 = xin_save[((size_t)b * steps + t) * 2 + 1];
    const float dai[3] = {da_r, da_z, da_n};    // grads of the input-side pre-activations
    const float dah[3] = {da_r, da_z, dghn};    // grads of the hidden-side pre-activations
    float dx0 = 0.f, dx1 = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int row = g * HID + j;
      P[OFF_WIH + row * 2] += dai[g] * xi0;
      P[OFF_WIH + row * 2 + 1] += dai[g] * xi1;
      P[OFF_BIH + row] += dai[g];
      P[OFF_BHH + row] += dah[g];
      dx0 += w_ih[row * 2] * dai[g];
      dx1 += w_ih[row * 2 + 1] * dai[g];
      sd[row] = dah[g];
    }
    sred[0][j] = dx0;
    sred[1][j] = dx1;
    shp[j] = hp;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int row = g * HID + j;
      float* wp = P + OFF_WHH + (size_t)row * HID;
      const float dg = dah[g];
      for (int k = 0; k < HID; ++k) wp[k] += dg * shp[k];
    }
    // dh_prev[j] += sum_rows W_hh[row][j] * dah[row]
    float acc = 0.f;
    for (int row = 0; row < 3 * HID; ++row) acc += w_hh[(size_t)row * HID + j] * sd[row];
    dh_prev += acc;
    float s0 = 0.f, s1 = 0.f;
    for (int k = 0; k < HID; ++k) { s0 += sred[0][k]; s1 += sred[1][k]; }
    __syncthreads();
    gh = dh_prev;
    gx0 += s0;  // xin_t = x_{t-1} + target, and x_t = x_{t-1} + ...
    gx1 += s1;
  }
  dz0[(size_t)b * HID + j] = gh;
}

}  // namespace

extern "C" int64_t mmfn_gru_head_part_floats(void) { return NPART; }

extern "C" int mmfn_gru_head_fwd_f32(const float* z0, const float* target, const float* w_ih, const float* w_hh,
                                     const float* b_ih, const float* b_hh, const float* w_out, const float* b_out,
                                     const float* gt, float* pred, float* hs, float* gates, float* xin, float* loss_terms,
                                     float* loss, int B, int steps, void* stream) {
  if (B <= 0 || steps <= 0 || steps > STEPS_MAX) return MMFN_EINVAL;
  if ((hs == nullptr) != (gates == nullptr) || (hs == nullptr) != (xin == nullptr)) return MMFN_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gru_head_fwd_kernel, dim3(B), dim3(HID), 0, s, z0, target, w_ih, w_hh, b_ih, b_hh, w_out, b_out, gt, pred, hs,
                     gates, xin, (gt && loss_terms) ? loss_terms : nullptr, steps);
  MMFN_LAUNCH_CHECK();
  if (gt && loss_terms && loss) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, loss_terms, B, 1.0f / (float)(B * steps * 2), loss);
    MMFN_LAUNCH_CHECK();
  }
  return 0;
}

// dpred == NULL: L1-loss gradient sign(pred-gt)*gscale (gscale = 1/(B*steps*2) for the mean);
// dpred != NULL: external gradient (autograd entry), scaled by gscale.
extern "C" int mmfn_gru_head_bwd_f32(const float* pred, const float* gt, const float* dpred, float gscale, const float* w_ih,
                                     const float* w_hh, const float* w_out, const float* hs, const float* gates,
                                     const float* xin, float* dz0, float* part, int B, int steps, void* stream) {
  if (B <= 0 || steps <= 0 || steps > STEPS_MAX || (!dpred && !gt)) return MMFN_EINVAL;
  hipLaunchKernelGGL(gru_head_bwd_kernel, dim3(B), dim3(HID), 0, (hipStream_t)stream, pred, gt, dpred, gscale, w_ih, w_hh, w_out,
                     hs, gates, xin, dz0, part, steps);
  MMFN_LAUNCH_CHECK();
  return 0;
}
