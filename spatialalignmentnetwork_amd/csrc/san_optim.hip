// Fused AdamW over one flat fp32 parameter buffer (gfx950).  Replaces the reference's
// torch.optim.AdamW over ~510 small tensors (model.py:72-87): the parameters, gradients and both
// moments of a network live in four flat buffers (dist.ParamBucket), so an optimisation step is ONE
// HBM-bound launch: 4 reads + 3 writes of 4 bytes per parameter (28 B), float4 accesses.
#include "san_common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    size_t count;
    float decay;        // 1 - lr * weight_decay
    float beta1, beta2;
    float step_size;    // lr / (1 - beta1^t)
    float inv_sqrt_bc2; // 1 / sqrt(1 - beta2^t)
    float eps;
    float grad_scale;   // e.g. 1 / world_size after a summing all-reduce
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
    g *= a.grad_scale;
    p *= a.decay;
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adamw_kernel(const AdamArgs a) {
    const size_t n4 = a.count / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 p = reinterpret_cast<f4*>(a.p)[i];
        const f4 g = reinterpret_cast<const f4*>(a.g)[i];
        f4 m = reinterpret_cast<f4*>(a.m)[i];
        f4 v = reinterpret_cast<f4*>(a.v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            adam_one(pk, g[k], mk, vk, a);
            p[k] = pk;
            m[k] = mk;
            v[k] = vk;
        }
        reinterpret_cast<f4*>(a.p)[i] = p;
        reinterpret_cast<f4*>(a.m)[i] = m;
        reinterpret_cast<f4*>(a.v)[i] = v;
    }
    // tail (count % 4 elements)
    const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < a.count) adam_one(a.p[t], a.g[t], a.m[t], a.v[t], a);
}

// The same step with the step count in DEVICE memory (hipGraph replays cannot change kernel arguments): every thread
// derives the bias corrections from *step + 1; adamw_count_kernel then advances the counter.
__global__ void __launch_bounds__(256) adamw_dev_kernel(AdamArgs a, float lr, const long long* __restrict__ step,
                                                        const float* __restrict__ hyper) {
    if (hyper) {            // (lr, weight_decay, grad_scale) in device memory: a captured launch follows a learning-rate schedule
        lr = hyper[0];
        a.decay = 1.f - lr * hyper[1];
        a.grad_scale = hyper[2];
    }
    const double t = (double)(step[0] + 1);
    a.step_size = (float)((double)lr / (1.0 - pow((double)a.beta1, t)));
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, t)));
    const size_t n4 = a.count / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 p = reinterpret_cast<f4*>(a.p)[i];
        const f4 g = reinterpret_cast<const f4*>(a.g)[i];
        f4 m = reinterpret_cast<f4*>(a.m)[i];
        f4 v = reinterpret_cast<f4*>(a.v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = m[k], vk = v[k];
            adam_one(pk, g[k], mk, vk, a);
            p[k] = pk;
            m[k] = mk;
            v[k] = vk;
        }
        reinterpret_cast<f4*>(a.p)[i] = p;
        reinterpret_cast<f4*>(a.m)[i] = m;
        reinterpret_cast<f4*>(a.v)[i] = v;
    }
    const size_t tl = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tl < a.count) adam_one(a.p[tl], a.g[tl], a.m[tl], a.v[tl], a);
}

__global__ void adamw_count_kernel(long long* step) { step[0] += 1; }

}  // namespace

extern "C" int san_adamw_step_dev(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, long long* step_dev, float grad_scale,
                                  void* stream) {
    return san_adamw_step_hyper(p, g, m, v, count, lr, beta1, beta2, eps, weight_decay, step_dev, grad_scale, nullptr, stream);
}

extern "C" int san_adamw_step_hyper(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, long long* step_dev, float grad_scale,
                                    const float* hyper_dev, void* stream) {
    SAN_CHECK_ARG(p && g && m && v && step_dev, "null pointer");
    SAN_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "buffers must be 16-byte aligned");
    AdamArgs a{};
    a.p = p;
    a.g = g;
    a.m = m;
    a.v = v;
    a.count = count;
    a.decay = 1.f - lr * weight_decay;
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.grad_scale = grad_scale;
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (count) hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, lr, step_dev, hyper_dev);
    hipLaunchKernelGGL(adamw_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}

extern "C" int san_adamw_step(float* p, const float* g, float* m, float* v, size_t count, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    SAN_CHECK_ARG(p && g && m && v, "null pointer");
    SAN_CHECK_ARG(step >= 1, "step counts from 1");
    SAN_CHECK_ARG(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "buffers must be 16-byte aligned");
    if (count == 0) return SAN_OK;
    AdamArgs a{};
    a.p = p;
    a.g = g;
    a.m = m;
    a.v = v;
    a.count = count;
    a.decay = 1.f - lr * weight_decay;
    a.beta1 = beta1;
    a.beta2 = beta2;
    // bias corrections in double on the host: the same numbers torch.optim.AdamW computes in Python floats
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.eps = eps;
    a.grad_scale = grad_scale;
    size_t blocks = (count / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    SAN_LAUNCH_CHECK();
    return SAN_OK;
}
