// The concatenating / softmax / extended styles of QCALayer (rumpy/SISR/models/attention_manipulators/architectures.py:41-136): the
// squeeze-excite vector of a block is a small MLP of the block's channel means AND the image's attribute vector.  Rarely used ablation
// styles: they run as separate launches around the residual block (block kernel with pool sums -> gate here -> rumpy_ca_scale_res_fwd;
// backward: rumpy_ca_bwd_reduce -> gate backward here -> rumpy_ca_bwd_apply), not inside the one-launch RCAB kernels, and in exact fp32.
//   layer l:  v = [previous output (n_prev) ; attributes (M) if cat] ; v = relu(v) if relu_in ; z = W v + b ; out = act(z)
//   act: 0 none, 1 ReLU, 2 sigmoid, 3 sigmoid followed by softmax over the outputs
//   max_concat  : [64+M -> cr, ReLU] [cr -> 64, sigmoid]                   (QCALayer.forward :116-117)
//   softmax     : the same with act 3 on the last layer                    (:125-127)
//   mini_concat : [64 -> cr, none] [relu_in, cr+M -> 64, sigmoid]          (:118-120: the ReLU of conv_du acts on the concatenated vector)
//   extended_attention : [64+M -> 32, ReLU] [32+M -> 16, ReLU] [16+M -> cr, ReLU] [cr -> 64, sigmoid]   (:121-124)
// One workgroup per image; every layer is at most (64 + 256) x 64 multiply-adds.
#include "common.hpp"

constexpr int QCA_MAXV = 64 + 256;         // longest layer input: channel means + a 256-entry degradation embedding
constexpr int QCA_THREADS = 128;

__device__ __forceinline__ int qca_out_offset(const rumpy_qca_args& a, int l) {      // offset of layer l's output in an image's `acts` row
  int off = a.C;
  for (int k = 0; k < l; ++k) off += a.layers[k].n_out;
  return off;
}
__device__ __forceinline__ float qca_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void __launch_bounds__(QCA_THREADS) qca_gate_fwd_kernel(rumpy_qca_args a) {
  __shared__ float v[QCA_MAXV], red[QCA_THREADS];
  const int n = blockIdx.x, tid = threadIdx.x;
  float* acts = a.acts + (size_t)n * RUMPY_QCA_ACT_STRIDE;
  // channel means from the conv launch's per-tile partial sums
  for (int c = tid; c < a.C; c += QCA_THREADS) {
    float s = 0.f;
    for (int t = 0; t < a.ntiles; ++t) s += a.pool[((size_t)n * a.ntiles + t) * a.C + c];
    acts[c] = s * a.inv_hw;
  }
  __syncthreads();
  int in_off = 0;
  for (int l = 0; l < a.nlayers; ++l) {
    const rumpy_qca_layer& ly = a.layers[l];
    const int nin = ly.n_prev + (ly.cat ? a.M : 0);
    for (int i = tid; i < nin; i += QCA_THREADS) {
      float x = (i < ly.n_prev) ? acts[in_off + i] : a.attr[(size_t)n * a.M + (i - ly.n_prev)];
      v[i] = ly.relu_in ? fmaxf(x, 0.f) : x;
    }
    __syncthreads();
    const int out_off = qca_out_offset(a, l);
    float z = 0.f;
    if (tid < ly.n_out) {
      z = ly.b[tid];
      const float* w = ly.w + (size_t)tid * nin;
      for (int i = 0; i < nin; ++i) z += w[i] * v[i];
      if (ly.act == 1) z = fmaxf(z, 0.f);
      else if (ly.act >= 2) z = qca_sigmoid(z);
      acts[out_off + tid] = z;                 // act 3: the sigmoid value is what the backward pass needs; the softmax goes to `gate`
    }
    if (ly.act == 3) {                          // softmax over the n_out sigmoid values (nn.Softmax(dim=1) on [N,C,1,1])
      red[tid] = (tid < ly.n_out) ? z : -1e30f;
      __syncthreads();
      for (int s = QCA_THREADS / 2; s >= 1; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
      const float mx = red[0];
      __syncthreads();
      const float e = (tid < ly.n_out) ? __expf(z - mx) : 0.f;
      red[tid] = e;
      __syncthreads();
      for (int s = QCA_THREADS / 2; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
      z = e / red[0];
    }
    if (l == a.nlayers - 1 && tid < ly.n_out) a.gate[(size_t)n * a.C + tid] = z;
    __syncthreads();
    in_off = out_off;
  }
}

// dgate = sum over the image of dy * t2 (rumpy_ca_bwd_reduce's partial sums) -> back through the layers: `delta` keeps every layer's
// pre-activation gradient for the parameter launch, dpool = d(mean) / HW for rumpy_ca_bwd_apply
__global__ void __launch_bounds__(QCA_THREADS) qca_gate_bwd_kernel(rumpy_qca_args a) {
  __shared__ float d[QCA_THREADS], red[QCA_THREADS];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* acts = a.acts + (size_t)n * RUMPY_QCA_ACT_STRIDE;
  float* delta = a.delta + (size_t)n * RUMPY_QCA_ACT_STRIDE;
  float g = 0.f;
  if (tid < a.C)
    for (int k = 0; k < a.nchunks; ++k) g += a.partial[((size_t)n * a.nchunks + k) * a.C + tid];
  for (int l = a.nlayers - 1; l >= 0; --l) {
    const rumpy_qca_layer& ly = a.layers[l];
    const int out_off = qca_out_offset(a, l);
    const int nin = ly.n_prev + (ly.cat ? a.M : 0);
    float dz = 0.f;
    if (ly.act == 3) {                          // out = softmax(s), s = sigmoid(z): ds = out * (g - sum(out * g))
      const float o = (tid < ly.n_out) ? a.gate[(size_t)n * a.C + tid] : 0.f;
      red[tid] = o * g;
      __syncthreads();
      for (int s = QCA_THREADS / 2; s >= 1; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
      g = o * (g - red[0]);
      __syncthreads();
    }
    if (tid < ly.n_out) {
      const float o = acts[out_off + tid];
      if (ly.act >= 2) dz = g * o * (1.f - o);
      else if (ly.act == 1) dz = (o > 0.f) ? g : 0.f;
      else dz = g;
      delta[out_off + tid] = dz;
    }
    d[tid] = dz;
    __syncthreads();
    // gradient w.r.t. the previous output part of the input (the attributes are data); relu_in: through the ReLU on the input vector
    const int in_off = (l == 0) ? 0 : qca_out_offset(a, l - 1);
    float gi = 0.f;
    if (tid < ly.n_prev) {
      for (int o = 0; o < ly.n_out; ++o) gi += ly.w[(size_t)o * nin + tid] * d[o];
      if (ly.relu_in && !(acts[in_off + tid] > 0.f)) gi = 0.f;
    }
    __syncthreads();
    g = gi;
  }
  if (tid < a.C) a.dpool[(size_t)n * a.C + tid] = g * a.inv_hw;
}

// parameter gradients of `nitems` gate MLPs in one launch: block (item, layer); gw[o][i] = scale * sum_n delta[n][o] * v[n][i], gb[o] = scale * sum_n delta[n][o]
__global__ void __launch_bounds__(256) qca_bwd_params_kernel(const rumpy_qca_args* __restrict__ items) {
  const rumpy_qca_args& a = items[blockIdx.x];
  const int l = blockIdx.y;
  if (l >= a.nlayers) return;
  const rumpy_qca_layer& ly = a.layers[l];
  const int out_off = qca_out_offset(a, l), in_off = (l == 0) ? 0 : qca_out_offset(a, l - 1);
  const int nin = ly.n_prev + (ly.cat ? a.M : 0);
  for (int e = threadIdx.x; e < ly.n_out * (nin + 1); e += 256) {
    const int o = e / (nin + 1), i = e - o * (nin + 1);
    float s = 0.f;
    for (int n = 0; n < a.N; ++n) {
      const float dz = a.delta[(size_t)n * RUMPY_QCA_ACT_STRIDE + out_off + o];
      float x = 1.f;
      if (i < nin) {
        x = (i < ly.n_prev) ? a.acts[(size_t)n * RUMPY_QCA_ACT_STRIDE + in_off + i] : a.attr[(size_t)n * a.M + (i - ly.n_prev)];
        if (ly.relu_in) x = fmaxf(x, 0.f);
      }
      s += dz * x;
    }
    if (i < nin) ly.gw[(size_t)o * nin + i] = s * a.scale;
    else ly.gb[o] = s * a.scale;
  }
}

static int qca_check(const rumpy_qca_args* p, const char* what) {
  if (!p || p->nlayers < 1 || p->nlayers > 4 || p->N <= 0 || p->C <= 0 || p->C > QCA_THREADS || p->M < 0 || p->M > 256 || !p->acts) {
    rumpy_set_error("%s: bad argument", what); return RUMPY_E_ARG; }
  int total = p->C;
  for (int l = 0; l < p->nlayers; ++l) {
    const rumpy_qca_layer& ly = p->layers[l];
    if (!ly.w || !ly.b || ly.n_out <= 0 || ly.n_out > QCA_THREADS || ly.n_prev != (l == 0 ? p->C : p->layers[l - 1].n_out) || ly.act < 0 || ly.act > 3 ||
        (ly.act == 3 && l != p->nlayers - 1) || (ly.cat && !p->attr)) { rumpy_set_error("%s: bad layer %d", what, l); return RUMPY_E_ARG; }
    total += ly.n_out;
  }
  if (p->layers[p->nlayers - 1].n_out != p->C || total > RUMPY_QCA_ACT_STRIDE) { rumpy_set_error("%s: the last layer must have C outputs; at most %d saved values per image", what, RUMPY_QCA_ACT_STRIDE); return RUMPY_E_ARG; }
  return RUMPY_OK;
}
extern "C" int rumpy_qca_gate_fwd(const rumpy_qca_args* p, void* stream) {
  if (int rc = qca_check(p, "rumpy_qca_gate_fwd")) return rc;
  if (!p->pool || !p->gate || p->ntiles <= 0) { rumpy_set_error("rumpy_qca_gate_fwd: null pointer"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(qca_gate_fwd_kernel, dim3(p->N), dim3(QCA_THREADS), 0, (hipStream_t)stream, *p);
  return rumpy_check_launch("rumpy_qca_gate_fwd");
}
extern "C" int rumpy_qca_gate_bwd(const rumpy_qca_args* p, void* stream) {
  if (int rc = qca_check(p, "rumpy_qca_gate_bwd")) return rc;
  if (!p->partial || !p->gate || !p->dpool || !p->delta || p->nchunks <= 0) { rumpy_set_error("rumpy_qca_gate_bwd: null pointer"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(qca_gate_bwd_kernel, dim3(p->N), dim3(QCA_THREADS), 0, (hipStream_t)stream, *p);
  return rumpy_check_launch("rumpy_qca_gate_bwd");
}
extern "C" int rumpy_qca_bwd_params(const rumpy_qca_args* items_device, int32_t nitems, void* stream) {
  if (!items_device || nitems <= 0) { rumpy_set_error("rumpy_qca_bwd_params: bad argument"); return RUMPY_E_ARG; }
  hipLaunchKernelGGL(qca_bwd_params_kernel, dim3(nitems, 4), dim3(256), 0, (hipStream_t)stream, items_device);
  return rumpy_check_launch("rumpy_qca_bwd_params");
}
