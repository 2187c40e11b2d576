// Reward normalisation with running return statistics, device resident.
//
// Replaces VecRunner::normalized_rewards (ppo/src/common/runner.rs:114-127) and
// RunningMeanStd<Ix0>::{new, update, update_from_moments} (ppo/src/common/running_mean_std.rs:16-67):
//   ret      = ret * gamma + rews                                   runner.rs:119
//   ret_rms.update(ret)   batch mean / population variance over the envs, Chan merge with
//                         count0 = 1e-4, mean0 = 0, var0 = 1          running_mean_std.rs:23-67
//   norm     = clamp(rews / sqrt(var + epsilon), -cliprew, cliprew)   runner.rs:121-122
//   ret      = ret * dones_mask     (sic: multiplies by `dones`, not `1 - dones`; reproduced)  runner.rs:124
// This is the step immediately BEFORE ExperienceBuffer::add_experience; with it a device-resident
// simulator can feed ppo_xp_add_device without a host round trip.
//
// Two launches per step: (1) ret update + fp64 (sum, sum of squares) per 256-env block,
// (2) every block folds the partials in the same fixed order, merges them into the running
// statistics with the reference's fp32 operation order, normalises its 256 rewards and masks its
// returns; block 0 stores the new running statistics into the OTHER of two state slots, so no
// block ever reads a slot that is being written.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../include/ppo_xp.h"
#include "errors.h"

namespace {

struct RmsState { float mean, var, count, pad; };

__device__ __forceinline__ double wsum(double x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  return x;
}

__global__ __launch_bounds__(256) void rnorm_accumulate_kernel(float *__restrict__ ret, const float *__restrict__ rews,
                                                               double2 *__restrict__ partials, unsigned N, float gamma) {
  __shared__ double red[4][2];
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  double r = 0.0;
  if (e < N) {
    const float x = ret[e] * gamma + rews[e];                         // runner.rs:119
    ret[e] = x;
    r = (double)x;
  }
  const double s1 = wsum(r), s2 = wsum(r * r);
  const unsigned w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) { red[w][0] = s1; red[w][1] = s2; }
  __syncthreads();
  if (threadIdx.x == 0)
    partials[blockIdx.x] = make_double2((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]),
                                        (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
}

// Fixed-order fold of the per-block partials by one 256-thread workgroup; the result is valid in
// thread 0.  Every workgroup that calls it gets the same bits.
__device__ __forceinline__ void fold_partials(const double2 *__restrict__ partials, unsigned nparts,
                                              double (*red)[2], double &t1, double &t2) {
  double s1 = 0.0, s2 = 0.0;
  for (unsigned i = threadIdx.x; i < nparts; i += 256u) { const double2 p = partials[i]; s1 += p.x; s2 += p.y; }
  s1 = wsum(s1); s2 = wsum(s2);
  const unsigned w = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0) { red[w][0] = s1; red[w][1] = s2; }
  __syncthreads();
  t1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
  t2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
}

// Sharded jobs: this rank's (count, sum, sum of squares) of the updated returns, for the all-reduce.
__global__ __launch_bounds__(256) void rnorm_fold_kernel(const double2 *__restrict__ partials, unsigned nparts,
                                                         double *__restrict__ sums, unsigned N) {
  __shared__ double red[4][2];
  double t1, t2;
  fold_partials(partials, nparts, red, t1, t2);
  if (threadIdx.x == 0) { sums[0] = (double)N; sums[1] = t1; sums[2] = t2; }
}

// global (optional, sharded jobs): (count, sum, sumsq) over ALL ranks' envs, already all-reduced; the
// local partials are then not read.
__global__ __launch_bounds__(256) void rnorm_apply_kernel(float *__restrict__ ret, const float *__restrict__ rews,
                                                          const float *__restrict__ dones, float *__restrict__ out,
                                                          const double2 *__restrict__ partials, unsigned nparts,
                                                          const RmsState *__restrict__ cur, RmsState *__restrict__ next,
                                                          const double *__restrict__ global, unsigned N, float epsilon,
                                                          float cliprew) {
  __shared__ double red[4][2];
  __shared__ float s_var;
  double t1 = 0.0, t2 = 0.0;
  if (!global) fold_partials(partials, nparts, red, t1, t2);   // kernel-uniform
  if (threadIdx.x == 0) {
    double n = (double)N;
    if (global) { n = global[0]; t1 = global[1]; t2 = global[2]; }
    const double bm = t1 / n;
    double bv = (t2 - t1 * bm) / n;
    bv = bv < 0.0 ? 0.0 : bv;                                          // cancellation only; NaN (inf or NaN returns) stays NaN like the reference's Welford
    // update_from_moments, running_mean_std.rs:45-67, fp32, operation order as written there
    const RmsState c = *cur;
    const float batch_mean = (float)bm, batch_var = (float)bv, b_count = (float)n;
    const float delta = batch_mean - c.mean;
    const float tot_count = c.count + b_count;
    const float new_mean = c.mean + delta * b_count / tot_count;
    const float m_a = c.var * c.count;
    const float m_b = batch_var * b_count;
    const float m2 = m_a + m_b + (delta * delta) * c.count * b_count / tot_count;
    const float new_var = m2 / tot_count;
    s_var = new_var;
    if (blockIdx.x == 0) { RmsState o; o.mean = new_mean; o.var = new_var; o.count = tot_count; o.pad = 0.f; *next = o; }
  }
  __syncthreads();
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e < N) {
    const float denom = sqrtf(s_var + epsilon);                        // runner.rs:121
    float x = rews[e] / denom;
    // runner.rs:122 f32::clamp: two ordered comparisons, so a NaN reward stays NaN (fminf/fmaxf
    // would turn it into -cliprew)
    if (x < -cliprew) x = -cliprew;
    if (x > cliprew) x = cliprew;
    out[e] = x;
    ret[e] = ret[e] * dones[e];                                        // runner.rs:124 (sic)
  }
}

}  // namespace

struct ppo_xp_rnorm {
  int device = 0;
  size_t N = 0;
  float gamma = 0.99f, epsilon = 1e-8f, cliprew = 10.f;
  float *ret = nullptr, *rews = nullptr, *dones = nullptr, *out = nullptr;   // [N]; rews/dones/out: staging for the host entry point
  double2 *partials = nullptr; unsigned nparts = 0;
  RmsState *state = nullptr;   // two slots
  double *sums = nullptr;      // [3] this rank's (count, sum, sumsq); all-reduced in place by the caller (sharded jobs)
  int cur = 0;
  bool pending = false;        // step_begin done, step_finish not yet
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;   // recorded after the last launch of every step, on whichever stream ran it
};

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return ppoxp_fail(_e == hipErrorOutOfMemory ? PPO_XP_ENOMEM : PPO_XP_EIO, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                          \
  } while (0)

extern "C" {

int ppo_xp_rnorm_create(ppo_xp_rnorm **out, int device, size_t num_envs, float gamma, float epsilon, float cliprew) {
  if (!out) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_create: out is NULL");
  *out = nullptr;
  if (num_envs == 0 || num_envs > 0x7fffffffull) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_create: bad num_envs");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return ppoxp_fail(PPO_XP_ENODEV, "rnorm_create: no HIP device %d (this library has no CPU path)", device);
  HIP_TRY(hipSetDevice(device));
  ppo_xp_rnorm *r = new (std::nothrow) ppo_xp_rnorm();
  if (!r) return ppoxp_fail(PPO_XP_ENOMEM, "rnorm_create: host allocation failed");
  r->device = device; r->N = num_envs; r->gamma = gamma; r->epsilon = epsilon; r->cliprew = cliprew;
  r->nparts = (unsigned)((num_envs + 255) / 256);
  const size_t fb = num_envs * sizeof(float);
  const RmsState init[2] = {{0.0f, 1.0f, 1e-4f, 0.f}, {0.0f, 1.0f, 1e-4f, 0.f}};   // running_mean_std.rs:20,29-33
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMalloc((void **)&r->ret, fb);
  if (e == hipSuccess) e = hipMalloc((void **)&r->rews, fb);
  if (e == hipSuccess) e = hipMalloc((void **)&r->dones, fb);
  if (e == hipSuccess) e = hipMalloc((void **)&r->out, fb);
  if (e == hipSuccess) e = hipMalloc((void **)&r->partials, r->nparts * sizeof(double2));
  if (e == hipSuccess) e = hipMalloc((void **)&r->state, sizeof(init));
  if (e == hipSuccess) e = hipMalloc((void **)&r->sums, 3 * sizeof(double));
  if (e == hipSuccess) e = hipMemset(r->ret, 0, fb);                                 // runner.rs:63
  if (e == hipSuccess) e = hipMemcpy(r->state, init, sizeof(init), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&r->done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) { ppo_xp_rnorm_destroy(r); return ppoxp_fail(PPO_XP_EIO, "rnorm_create: %s", hipGetErrorString(e)); }
  *out = r;
  return 0;
}

void ppo_xp_rnorm_destroy(ppo_xp_rnorm *r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->done) { (void)hipEventSynchronize(r->done); (void)hipEventDestroy(r->done); }
  if (r->stream) { (void)hipStreamSynchronize(r->stream); (void)hipStreamDestroy(r->stream); }
  (void)hipFree(r->ret); (void)hipFree(r->rews); (void)hipFree(r->dones); (void)hipFree(r->out);
  (void)hipFree(r->partials); (void)hipFree(r->state); (void)hipFree(r->sums);
  delete r;
}

int ppo_xp_rnorm_step_device(ppo_xp_rnorm *r, const float *rewards_dev, const float *dones_f32_dev, float *out_dev,
                             size_t n, void *stream) {
  if (!r || !rewards_dev || !dones_f32_dev || !out_dev) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step: NULL argument");
  if (n != r->N) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step: %zu rewards for %zu envs", n, r->N);
  if (r->pending) return ppoxp_fail(PPO_XP_ESTATE, "rnorm_step: a sharded step is open (call ppo_xp_rnorm_step_finish_device)");
  HIP_TRY(hipSetDevice(r->device));
  hipStream_t s = stream ? (hipStream_t)stream : r->stream;
  const unsigned N = (unsigned)r->N;
  hipLaunchKernelGGL(rnorm_accumulate_kernel, dim3(r->nparts), dim3(256), 0, s, r->ret, rewards_dev, r->partials, N, r->gamma);
  hipLaunchKernelGGL(rnorm_apply_kernel, dim3(r->nparts), dim3(256), 0, s, r->ret, rewards_dev, dones_f32_dev, out_dev,
                     r->partials, r->nparts, r->state + r->cur, r->state + (r->cur ^ 1), (const double *)nullptr, N,
                     r->epsilon, r->cliprew);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ppoxp_fail(PPO_XP_EIO, "rnorm_step: launch failed: %s", hipGetErrorString(e));
  HIP_TRY(hipEventRecord(r->done, s));
  r->cur ^= 1;
  return 0;
}

int ppo_xp_rnorm_step_begin_device(ppo_xp_rnorm *r, const float *rewards_dev, size_t n, void *stream) {
  if (!r || !rewards_dev) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step_begin: NULL argument");
  if (n != r->N) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step_begin: %zu rewards for %zu envs", n, r->N);
  if (r->pending) return ppoxp_fail(PPO_XP_ESTATE, "rnorm_step_begin: the previous sharded step was not finished");
  HIP_TRY(hipSetDevice(r->device));
  hipStream_t s = stream ? (hipStream_t)stream : r->stream;
  const unsigned N = (unsigned)r->N;
  hipLaunchKernelGGL(rnorm_accumulate_kernel, dim3(r->nparts), dim3(256), 0, s, r->ret, rewards_dev, r->partials, N, r->gamma);
  hipLaunchKernelGGL(rnorm_fold_kernel, dim3(1), dim3(256), 0, s, r->partials, r->nparts, r->sums, N);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ppoxp_fail(PPO_XP_EIO, "rnorm_step_begin: launch failed: %s", hipGetErrorString(e));
  HIP_TRY(hipEventRecord(r->done, s));
  r->pending = true;
  return 0;
}

double *ppo_xp_rnorm_sums_device(ppo_xp_rnorm *r) { return r ? r->sums : nullptr; }

int ppo_xp_rnorm_step_finish_device(ppo_xp_rnorm *r, const float *rewards_dev, const float *dones_f32_dev, float *out_dev,
                                    size_t n, void *stream) {
  if (!r || !rewards_dev || !dones_f32_dev || !out_dev) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step_finish: NULL argument");
  if (n != r->N) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step_finish: %zu rewards for %zu envs", n, r->N);
  if (!r->pending) return ppoxp_fail(PPO_XP_ESTATE, "rnorm_step_finish: no sharded step is open");
  HIP_TRY(hipSetDevice(r->device));
  hipStream_t s = stream ? (hipStream_t)stream : r->stream;
  hipLaunchKernelGGL(rnorm_apply_kernel, dim3(r->nparts), dim3(256), 0, s, r->ret, rewards_dev, dones_f32_dev, out_dev,
                     r->partials, r->nparts, r->state + r->cur, r->state + (r->cur ^ 1), (const double *)r->sums,
                     (unsigned)r->N, r->epsilon, r->cliprew);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ppoxp_fail(PPO_XP_EIO, "rnorm_step_finish: launch failed: %s", hipGetErrorString(e));
  HIP_TRY(hipEventRecord(r->done, s));
  r->cur ^= 1;
  r->pending = false;
  return 0;
}

int ppo_xp_rnorm_step(ppo_xp_rnorm *r, const float *rewards, const uint8_t *dones, float *out, size_t n) {
  if (!r || !rewards || !dones || !out) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step: NULL argument");
  if (n != r->N) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_step: %zu rewards for %zu envs", n, r->N);
  HIP_TRY(hipSetDevice(r->device));
  std::vector<float> df(n);
  for (size_t i = 0; i < n; ++i) df[i] = (float)(uint8_t)(dones[i] != 0);           // runner.rs:116
  HIP_TRY(hipMemcpyAsync(r->rews, rewards, n * sizeof(float), hipMemcpyHostToDevice, r->stream));
  HIP_TRY(hipMemcpyAsync(r->dones, df.data(), n * sizeof(float), hipMemcpyHostToDevice, r->stream));
  if (int rc = ppo_xp_rnorm_step_device(r, r->rews, r->dones, r->out, n, nullptr)) return rc;
  HIP_TRY(hipMemcpyAsync(out, r->out, n * sizeof(float), hipMemcpyDeviceToHost, r->stream));
  HIP_TRY(hipStreamSynchronize(r->stream));
  return 0;
}

int ppo_xp_rnorm_state(ppo_xp_rnorm *r, float *mean, float *var, float *count, float *ret_out, size_t n) {
  if (!r) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_state: handle is NULL");
  HIP_TRY(hipSetDevice(r->device));
  HIP_TRY(hipEventSynchronize(r->done));   // a caller's stream may have run the last step
  HIP_TRY(hipStreamSynchronize(r->stream));
  RmsState st;
  HIP_TRY(hipMemcpy(&st, r->state + r->cur, sizeof(st), hipMemcpyDeviceToHost));
  if (mean) *mean = st.mean;
  if (var) *var = st.var;
  if (count) *count = st.count;
  if (ret_out) {
    if (n != r->N) return ppoxp_fail(PPO_XP_EINVAL, "rnorm_state: n = %zu, envs = %zu", n, r->N);
    HIP_TRY(hipMemcpy(ret_out, r->ret, n * sizeof(float), hipMemcpyDeviceToHost));
  }
  return 0;
}

}  // extern "C"
