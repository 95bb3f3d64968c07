// Handle-level C ABI: the whole forward of the canonical U-Net (unet.py:65-69 over SURVEY.md 8d's network) as ONE call, for a caller
// that is not Python -- create a net from its configuration, hand it the fp32 parameters as one flat device vector (canonical order,
// iunet_net_param), let it fold / scale / pack them into a caller-owned device buffer, then run forwards on a caller-owned workspace.
// The launch graph that interactive_unet/engine.py and engine_x2.py sequence from Python is sequenced here in C++ from the same entry
// points.  Modes: 2 = fp16x2 split precision (the tolerance-meeting default: logits within 1e-3 of the fp32 reference predict,
// predict.py:30-35), 3 = the same with the cross terms of the 3-D stage convs on the fp8 matrix cores (conv3_x2m.hip: what the 3-D
// prediction runs), 0 / 1 = fp16 / bf16 activations (the throughput path).  No device allocation, no synchronisation: the handle is
// host memory only.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int iunet_x2_prep(const void*, void*, void*, void*, const void*, const void*, const void*, const void*, const void*, float, float, float,
                  int, int, int, int, int, void*);
int iunet_x2_first_conv_fwd(int, const void*, int, const long long*, void*, long long, int, const void*, const void*, const void*, float,
                            int, int, int, int, int, int, int, void*);
int iunet_x2_conv3_fwd(int, const void*, long long, int, void*, long long, int, const void*, const void*, const void*, int, int, int, int,
                       int, int, int, void*);
int iunet_x2_maxpool_fwd(int, const void*, long long, int, void*, long long, int, int, int, int, int, int, void*);
int iunet_x2_convT_fwd(int, const void*, long long, int, void*, long long, int, const void*, const void*, const void*, int, int, int, int,
                       int, int, void*);
int iunet_x2_head_fwd(const void*, long long, int, int, const void*, const void*, float, int, void*, void*, void*, const long long*, float,
                      int, int, int, int, int, void*);
int iunet_x2m_prep_nd(int, const void*, void*, void*, void*, void*, const void*, const void*, const void*, const void*, float, float, float, int, int, void*);
int iunet_x2m_conv_fwd(int, const void*, long long, const void*, long long, void*, long long, int, void*, long long, const void*, const void*,
                       const void*, const void*, int, int, int, int, int, int, int, void*, void*);
long long iunet_x2m_w8_bytes_nd(int, int, int);
int iunet_x2m_head_fusable(int, int);
int iunet_x2m_conv_head_fwd(int, const void*, long long, const void*, long long, const void*, const void*, const void*, const void*, const void*,
                            const void*, float, int, void*, void*, void*, const long long*, float, int, int, int, int, int, int, void*, void*);
int iunet_x2m_first_conv_fwd(int, const void*, int, const long long*, void*, long long, int, void*, long long, const void*, const void*, const void*,
                             float, int, int, int, int, int, int, int, void*, void*);
int iunet_x2m_convT_fwd(int, const void*, long long, int, void*, long long, int, void*, long long, const void*, const void*, const void*, int, int,
                        int, int, int, int, void*, void*);
int iunet_x2_conv3_fwd_flag(int, const void*, long long, int, void*, long long, int, const void*, const void*, const void*, int, int, int, int,
                            int, int, int, void*, void*);
int iunet_x2m_maxpool_fwd(int, const void*, long long, const void*, long long, void*, long long, void*, long long, int, int, int, int, int, void*);
int iunet_x2m_pool_fusable(int, int);
int iunet_x2m_first_stage_fusable(int, int, int, int, int, int);
int iunet_x2m_first_stage_fwd(const void*, int, const long long*, const void*, const void*, const void*, float, void*, long long, int, void*, long long, void*,
                              long long, void*, long long, const void*, const void*, const void*, const void*, int, int, int, void*, void*);
int iunet_x2m_conv_pool_fwd(int, const void*, long long, const void*, long long, void*, long long, int, void*, long long, void*, long long, void*, long long,
                            const void*, const void*, const void*, const void*, int, int, int, int, int, int, int, void*, void*);
int iunet_x2m_conv3_fwd(const void*, long long, const void*, long long, void*, long long, int, void*, long long, const void*, const void*,
                        const void*, const void*, int, int, int, int, int, int, int, void*, void*);
long long iunet_x2m_w8_bytes(int, int);
int iunet_pack_conv3(int, const void*, const void*, void*, int, int, int, int, void*);
int iunet_pack_first_conv(int, const void*, const void*, void*, int, int, int, void*);
int iunet_pack_convT(int, const void*, void*, int, int, int, void*);
long long iunet_pack_conv3_elems(int, int, int, int);
long long iunet_pack_first_conv_elems(int, int, int);
int iunet_conv3_pick_layout(int, int, int, int, int, int, int);
int iunet_x2_pack_mode(int);
int iunet_conv3_compact_ok(int, int, int, int, int, int, int, int, int);
int iunet_first_conv_fwd(int, int, const void*, int, const long long*, void*, long long, const void*, const void*, void*, int, int, int,
                         int, int, int, int, void*);
int iunet_conv3_fwd(int, int, const void*, long long, void*, long long, const void*, const void*, void*, int, int, int, int, int, int,
                    int, int, void*);
int iunet_maxpool_fwd(int, int, const void*, long long, void*, long long, int, int, int, int, int, void*);
int iunet_convT_fwd(int, int, const void*, long long, void*, long long, const void*, const void*, int, int, int, int, int, int, void*);
long long iunet_gn_precise_slab_bytes(int, int, long long);
int iunet_x2_gn_relu_fwd(const void*, long long, int, void*, long long, int, const void*, const void*, int, float, float, void*, void*, void*, int, int,
                         long long, void*, void*);
int iunet_head_loss_num_parts(int, long long);
int iunet_head_loss_fwd(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, int, void*, void*,
                        void*, int, long long, void*);
int iunet_head_fwd(int, const void*, long long, int, const void*, const void*, int, void*, void*, void*, const long long*, float, int, int,
                   int, int, int, void*);
}

namespace {

// eval-mode BatchNorm fold of the 16-bit modes: scale = gamma / sqrt(var + eps), bias = beta - mean * scale, every operation rounded
// on its own (pack_batch.hip: fold_scale / fold_bias; oracle/unet_ref.py: fold_bn)
__global__ void net_fold_bn_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                   const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ bias, int C) {
#pragma clang fp contract(off)
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float s = var[c] + eps;
  const float a = gamma[c] / sqrtf(s);
  const float t = mean[c] * a;
  scale[c] = a;
  bias[c] = beta[c] - t;
}

long long align256(long long v) { return (v + 255) & ~255ll; }

struct Param { std::string name; long long off, numel; };

struct ConvOp {                       // one stage conv (or the first conv): where its parameters and packed operators live
  int ci, co, first;
  long long w, bn;                    // flat-parameter offsets: weight; gamma (beta, mean, var follow, co each)
  long long pk[4];                    // packed-buffer byte offsets by layout (0, 1 = K16, 3 = compact; -1 = absent); x2: pk[1]; x2m: pk[1] = w_hi (K16), pk[0] = K128 bytes
  long long aux;                      // x2: [oscale co | bias co] floats; 16-bit: [scale co | bias co]
};
struct UpOp { int ci, co; long long w, b, pk, aux; };

}  // namespace

struct iunet_net {
  int dim, levels, base, cin, ncls, mode;
  int norm = 0, groups = 8;           // norm 1: GroupNorm(groups) + ReLU after every stage conv (mode 2 only): nothing folds, csrc/gn_precise.hip normalises
  float act_scale;
  int taps, npos;
  std::vector<int> ch;
  std::vector<Param> params;
  long long nparams = 0;
  std::vector<ConvOp> conv;           // enc0.conv1, enc0.conv2, ..., dec{L-2}.conv1, ... in stage order
  std::vector<UpOp> up;               // dec{L-2}.up ... dec0.up
  long long head_w = 0, head_b = 0;
  long long packed_bytes = 0, scratch_off = 0;
  const float* flat = nullptr;        // set by iunet_net_load
  unsigned char* packed = nullptr;
};

namespace {

int stage_index(const iunet_net* n, bool dec, int l) { return dec ? n->levels + (n->levels - 2 - l) : l; }

struct WsLayout { std::vector<long long> a, b, cat, pin, am, catm, pinm; long long raw = -1, gnslab = -1, gnsc = -1, gnsh = -1; long long bytes; };

// activation buffers of one forward (elements of 2 bytes; x2 holds hi + lo planes: twice the channels).  Modes 2 and 3: the first 256
// bytes hold the range flag (an int the forward raises to 0x7bff when a stored hi word saturates; the caller zeroes it once).  x2m (mode 3):
// a / cat / pin are hi planes + lo8 planes (2 + 1 bytes per element), b hi + lo planes
WsLayout ws_layout(const iunet_net* n, int N, int D, int H, int W) {
  WsLayout L;
  const int lv = n->levels, mul = n->mode == 2 ? 2 : 1;
  long long off = n->mode >= 2 ? 256 : 0;
  auto take = [&](long long elems) { const long long o = off; off = align256(off + elems * 2); return o; };
  L.a.resize(lv); L.b.resize(lv); L.cat.resize(lv, -1); L.pin.resize(lv, -1);
  L.am.resize(lv, -1); L.catm.resize(lv, -1); L.pinm.resize(lv, -1);
  if (n->mode == 3) {
    for (int l = 0; l < lv; ++l) {
      const long long v = (long long)(n->dim == 3 ? D >> l : 1) * (H >> l) * (W >> l);
      L.a[l] = take((long long)N * n->ch[l] * v); L.am[l] = take((long long)N * n->ch[l] * v / 2);
      L.b[l] = take((long long)N * 2 * n->ch[l] * v);
      if (l < lv - 1) { L.cat[l] = take((long long)N * 2 * n->ch[l] * v); L.catm[l] = take((long long)N * n->ch[l] * v); }
      if (l > 0) { L.pin[l] = take((long long)N * n->ch[l - 1] * v); L.pinm[l] = take((long long)N * n->ch[l - 1] * v / 2); }
    }
    L.bytes = off;
    return L;
  }
  for (int l = 0; l < lv; ++l) {
    const long long v = (long long)(n->dim == 3 ? D >> l : 1) * (H >> l) * (W >> l);
    L.a[l] = take((long long)N * mul * n->ch[l] * v);
    L.b[l] = take((long long)N * mul * n->ch[l] * v);
    if (l < lv - 1) L.cat[l] = take((long long)N * mul * 2 * n->ch[l] * v);
    if (l > 0) L.pin[l] = take((long long)N * mul * n->ch[l - 1] * v);
  }
  if (n->norm == 1) {        // GroupNorm: the raw output of the conv in flight, the statistics slab, the per-sample affine pairs
    long long mx = 0, slab = 0;
    for (int l = 0; l < lv; ++l) {
      const long long v = (long long)(n->dim == 3 ? D >> l : 1) * (H >> l) * (W >> l);
      mx = std::max(mx, (long long)n->ch[l] * v);
      slab = std::max(slab, iunet_gn_precise_slab_bytes(N, n->ch[l], v));
    }
    L.raw = take((long long)N * 2 * mx);
    L.gnslab = take(slab / 2 + 1);
    L.gnsc = take(2ll * N * n->ch[lv - 1]);
    L.gnsh = take(2ll * N * n->ch[lv - 1]);
  }
  L.bytes = off;
  return L;
}

}  // namespace

extern "C" {

/* mode: 0 fp16, 1 bf16, 2 fp16x2 (split precision), 3 fp16x2 with the cross terms of the stage convs on the fp8 matrix cores (3-D only);
 * act_scale: power of two (modes 2, 3; 0 = the default 64) */
int iunet_net_create_ex(int dim, int levels, int base, int cin, int ncls, int mode, float act_scale, int norm, int groups, iunet_net** out);
int iunet_net_create(int dim, int levels, int base, int cin, int ncls, int mode, float act_scale, iunet_net** out) {
  return iunet_net_create_ex(dim, levels, base, cin, ncls, mode, act_scale, 0, 8, out);
}
/* norm: 0 BatchNorm (eval-mode statistics folded into the operators), 1 GroupNorm(groups) + ReLU after every stage conv (north star
 * "GroupNorm/BN"; mode 2, the split-precision form GroupNorm networks predict in: engine_x2.EngineX2(norm='group')) */
int iunet_net_create_ex(int dim, int levels, int base, int cin, int ncls, int mode, float act_scale, int norm, int groups, iunet_net** out) {
  IUNET_REQUIRE(out != nullptr, "net_create: null handle pointer");
  IUNET_REQUIRE(norm == 0 || norm == 1, "net_create: norm must be 0 (batch) or 1 (group), got %d", norm);
  IUNET_REQUIRE(norm == 0 || (mode == 2 && groups > 0 && base % groups == 0), "net_create: GroupNorm runs in mode 2 (fp16x2) with groups dividing base (mode %d, %d groups)", mode, groups);
  IUNET_REQUIRE(dim == 2 || dim == 3, "net_create: dim must be 2 or 3 (got %d)", dim);
  IUNET_REQUIRE(levels >= 2 && levels <= 6, "net_create: levels must be 2..6 (got %d)", levels);
  IUNET_REQUIRE(base > 0 && base % 32 == 0, "net_create: base channels must be a positive multiple of 32 (got %d)", base);
  IUNET_REQUIRE(cin >= 1 && cin <= 4, "net_create: 1..4 input channels (got %d)", cin);
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "net_create: 2..10 classes (app.py:162; got %d)", ncls);
  IUNET_REQUIRE(mode >= 0 && mode <= 3, "net_create: mode must be 0 (fp16), 1 (bf16), 2 (fp16x2) or 3 (fp16x2, cross terms on fp8), got %d", mode);
  iunet_net* n = new iunet_net();
  n->dim = dim; n->levels = levels; n->base = base; n->cin = cin; n->ncls = ncls; n->mode = mode;
  n->norm = norm; n->groups = groups;
  n->act_scale = act_scale > 0.f ? act_scale : 64.0f;
  n->taps = dim == 3 ? 27 : 9; n->npos = dim == 3 ? 8 : 4;
  for (int l = 0; l < levels; ++l) n->ch.push_back(base << l);
  long long off = 0, pk = 0;
  auto add = [&](const std::string& name, long long numel) { n->params.push_back({name, off, numel}); const long long o = off; off += numel; return o; };
  auto pk_take = [&](long long bytes) { const long long o = pk; pk = align256(pk + bytes); return o; };
  long long max_virtual = 0;
  auto stage = [&](const std::string& prefix, int ci, int co) {
    for (int j = 1; j <= 2; ++j) {
      ConvOp op;
      op.ci = j == 1 ? ci : co; op.co = co; op.first = (prefix == "enc0" && j == 1);
      const std::string c = prefix + ".conv" + std::to_string(j), b = prefix + ".bn" + std::to_string(j);
      op.w = add(c + ".weight", (long long)co * op.ci * n->taps);
      op.bn = add(b + ".weight", co); add(b + ".bias", co); add(b + ".running_mean", co); add(b + ".running_var", co);
      for (int k = 0; k < 4; ++k) op.pk[k] = -1;
      const int vci = (mode == 2 || (mode == 3 && op.first)) ? 3 * op.ci : op.ci;
      if (op.first) op.pk[1] = pk_take(iunet_pack_first_conv_elems(co, vci, n->taps) * 2);
      else if (mode == 3) {
        op.pk[1] = pk_take(iunet_pack_conv3_elems(co, op.ci, n->taps, dim == 3 ? 2 : 6) * 2);
        op.pk[0] = pk_take(iunet_x2m_w8_bytes_nd(dim, co, op.ci));
      } else {
        op.pk[1] = pk_take(iunet_pack_conv3_elems(co, vci, n->taps, mode == 2 ? iunet_x2_pack_mode(dim) : 2) * 2);
        if (mode != 2) {          // the layouts a 16-bit launch may pick (interactive_unet/_native.py: PackedConv)
          const bool compact2d = n->taps == 9 && iunet_conv3_compact_ok(2, 1, 1, 16, 32, op.ci, co, 0, 0);      // (off: IUNET_NO_COMPACT2D)
          if (co % 64 == 0 && n->taps == 9 && op.ci > 64 && !compact2d) op.pk[0] = pk_take(iunet_pack_conv3_elems(co, op.ci, n->taps, 0) * 2);
          const char* nc = getenv("IUNET_NO_COMPACT");                                  // (the switch _native.PackedConv honours: ADVICE r3)
          if (((n->taps == 27 && op.ci > 32) || compact2d) && !(nc && nc[0])) op.pk[3] = pk_take(iunet_pack_conv3_elems(co, op.ci, n->taps, 6) * 2);
        }
      }
      op.aux = pk_take(2ll * co * 4);
      if (mode >= 2 && 3ll * co * op.ci * n->taps > max_virtual) max_virtual = 3ll * co * op.ci * n->taps;
      n->conv.push_back(op);
    }
  };
  for (int l = 0; l < levels; ++l) stage("enc" + std::to_string(l), l == 0 ? cin : n->ch[l - 1], n->ch[l]);
  for (int l = levels - 2; l >= 0; --l) {
    UpOp u;
    u.ci = n->ch[l + 1]; u.co = n->ch[l];
    const std::string p = "dec" + std::to_string(l);
    u.w = add(p + ".up.weight", (long long)u.ci * u.co * n->npos);
    u.b = add(p + ".up.bias", u.co);
    u.pk = pk_take((long long)(mode >= 2 ? 2 : 1) * u.ci * u.co * n->npos * 2);
    u.aux = pk_take(2ll * u.co * 4);
    if (mode >= 2 && 2ll * u.ci * u.co * n->npos > max_virtual) max_virtual = 2ll * u.ci * u.co * n->npos;
    n->up.push_back(u);
    stage(p, 2 * n->ch[l], n->ch[l]);
  }
  n->head_w = add("head.weight", (long long)ncls * n->ch[0]);
  n->head_b = add("head.bias", ncls);
  n->nparams = off;
  n->scratch_off = pk;
  n->packed_bytes = pk + max_virtual * 4;          // x2: the virtual fp32 operator of the layer being prepared
  *out = n;
  return IUNET_OK;
}

void iunet_net_destroy(iunet_net* n) { delete n; }

/* fp32 elements of the flat parameter vector (trainable parameters AND BatchNorm running statistics, canonical order) */
long long iunet_net_num_params(const iunet_net* n) { return n ? n->nparams : 0; }
int iunet_net_num_tensors(const iunet_net* n) { return n ? (int)n->params.size() : 0; }

/* tensor `index` of the flat vector: its canonical name (the state_dict key of interactive_unet.unet.UNet / oracle/unet_ref.py), offset
 * and element count; shapes: conv [Cout][Cin][3^d], up [Cin][Cout][2^d], head [ncls][base] */
int iunet_net_param(const iunet_net* n, int index, char* name, int name_cap, long long* offset, long long* numel) {
  IUNET_REQUIRE(n != nullptr, "net_param: null handle");
  IUNET_REQUIRE(index >= 0 && index < (int)n->params.size(), "net_param: index %d out of range", index);
  const Param& p = n->params[index];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", p.name.c_str());
  if (offset) *offset = p.off;
  if (numel) *numel = p.numel;
  return IUNET_OK;
}

/* device bytes of the caller-owned buffer iunet_net_load writes the packed operators into */
long long iunet_net_packed_bytes(const iunet_net* n) { return n ? n->packed_bytes : 0; }

/* fold eval-mode BatchNorm, scale / split (mode 2) and reorder every operator: flat_params (device, iunet_net_num_params floats) ->
 * packed (device, iunet_net_packed_bytes).  Both buffers must stay valid (and flat_params unchanged, for the head) while forwards run;
 * call again after the parameters changed. */
int iunet_net_load(iunet_net* n, const void* flat_params, void* packed, void* stream) {
  IUNET_REQUIRE(n && flat_params && packed, "net_load: null pointer");
  const float* P = (const float*)flat_params;
  unsigned char* K = (unsigned char*)packed;
  const float eps = 1e-5f, A = n->act_scale;
  for (const ConvOp& op : n->conv) {
    const float* w = P + op.w;
    const float *g = P + op.bn, *be = g + op.co, *mu = be + op.co, *va = mu + op.co;
    float* aux = (float*)(K + op.aux);
    int rc;
    if (n->mode == 3 && !op.first) {
      float* whi = (float*)(K + n->scratch_off);
      IUNET_CHECK_HIP(hipMemsetAsync(K + op.pk[0], 0, (size_t)iunet_x2m_w8_bytes_nd(n->dim, op.co, op.ci), (hipStream_t)stream));
      rc = iunet_x2m_prep_nd(n->dim, w, whi, K + op.pk[0], aux, aux + op.co, g, be, mu, va, eps, A, A, op.co, op.ci, stream);
      if (rc) return rc;
      rc = iunet_pack_conv3(0, whi, nullptr, K + op.pk[1], op.co, op.ci, n->taps, n->dim == 3 ? 2 : 6, stream);
      if (rc) return rc;
    } else if (n->mode >= 2) {
      float* wv = (float*)(K + n->scratch_off);
      const bool gn = n->norm == 1;         // GroupNorm: the raw operator (gamma / beta go to the normalisation pass)
      rc = iunet_x2_prep(w, wv, aux, aux + op.co, gn ? nullptr : g, gn ? nullptr : be, gn ? nullptr : mu, gn ? nullptr : va, nullptr, eps, A, A,
                         op.co, op.ci, n->taps, 0, op.first ? op.ci : (n->dim == 3 ? 16 : 32), stream);
      if (rc) return rc;
      rc = op.first ? iunet_pack_first_conv(0, wv, nullptr, K + op.pk[1], op.co, 3 * op.ci, n->taps, stream)
                    : iunet_pack_conv3(0, wv, nullptr, K + op.pk[1], op.co, 3 * op.ci, n->taps, iunet_x2_pack_mode(n->dim), stream);
      if (rc) return rc;
    } else {
      hipLaunchKernelGGL(net_fold_bn_kernel, dim3((op.co + 255) / 256), dim3(256), 0, (hipStream_t)stream, g, be, mu, va, eps, aux,
                         aux + op.co, op.co);
      IUNET_CHECK_HIP(hipGetLastError());
      if (op.first) rc = iunet_pack_first_conv(n->mode, w, aux, K + op.pk[1], op.co, op.ci, n->taps, stream);
      else {
        rc = iunet_pack_conv3(n->mode, w, aux, K + op.pk[1], op.co, op.ci, n->taps, 2, stream);
        if (!rc && op.pk[0] >= 0) rc = iunet_pack_conv3(n->mode, w, aux, K + op.pk[0], op.co, op.ci, n->taps, 0, stream);
        if (!rc && op.pk[3] >= 0) rc = iunet_pack_conv3(n->mode, w, aux, K + op.pk[3], op.co, op.ci, n->taps, 6, stream);
      }
      if (rc) return rc;
    }
  }
  for (const UpOp& u : n->up) {
    float* aux = (float*)(K + u.aux);
    int rc;
    if (n->mode >= 2) {
      float* wv = (float*)(K + n->scratch_off);
      rc = iunet_x2_prep(P + u.w, wv, aux, aux + u.co, nullptr, nullptr, nullptr, nullptr, P + u.b, eps, A, A, u.co, u.ci, n->npos, 2, 0, stream);
      if (!rc) rc = iunet_pack_convT(0, wv, K + u.pk, 2 * u.ci, u.co, n->npos, stream);
    } else {
      rc = iunet_pack_convT(n->mode, P + u.w, K + u.pk, u.ci, u.co, n->npos, stream);
    }
    if (rc) return rc;
  }
  n->flat = P;
  n->packed = K;
  return IUNET_OK;
}

/* device bytes of the activation workspace of one forward of N samples on a D x H x W grid (D = 1 in 2-D); 0 on a bad shape */
long long iunet_net_workspace_bytes(const iunet_net* n, int N, int D, int H, int W) {
  if (!n || N < 1 || D < 1 || H < 1 || W < 1) return 0;
  const int f = 1 << (n->levels - 1);
  if (H % f || W % f || (n->dim == 3 && D % f) || (n->dim == 2 && D != 1)) return 0;
  return ws_layout(n, N, D, H, W).bytes;
}

/* unet.py:65-69 (+ predict.py:38's class map): x = the caller's tensor (in_dtype 0 f32, 1 f16, 2 u8 scaled by 1 / 255, 3 bf16; element
 * strides n, c, d, h, w) -> any of logits / probs (fp32, element strides out_strides n, c, d, h, w; probs: out = ((accumulate ? out : 0)
 * + p) / divisor, predict.py:101-110) and cls (uint8 [N][D*H*W]).  workspace: iunet_net_workspace_bytes. */
int iunet_net_forward(iunet_net* n, const void* x, int in_dtype, const long long* in_strides, int N, int D, int H, int W, void* workspace,
                      void* logits, void* probs, void* cls, const long long* out_strides, float divisor, int accumulate, void* stream) {
  IUNET_REQUIRE(n && x && in_strides && workspace, "net_forward: null pointer");
  IUNET_REQUIRE(n->packed != nullptr, "net_forward: iunet_net_load has not been called");
  IUNET_REQUIRE(iunet_net_workspace_bytes(n, N, D, H, W) > 0, "net_forward: spatial size %d x %d x %d must be divisible by %d (D == 1 in 2-D)",
                D, H, W, 1 << (n->levels - 1));
  IUNET_REQUIRE((logits == nullptr && probs == nullptr) || out_strides != nullptr, "net_forward: logits / probs need out_strides");
  const WsLayout L = ws_layout(n, N, D, H, W);
  unsigned char* WS = (unsigned char*)workspace;
  unsigned char* K = n->packed;
  const int lv = n->levels, dim = n->dim, mode = n->mode;
  if (mode == 3) {
    // ---- x2m: a / cat / pin = (hi planes, lo8 planes), b = (hi planes, lo planes); engine_x2.EngineX2._infer_mixed sequences the same launches
    const float A = n->act_scale;
    void* sat = WS;
    auto dims3 = [&](int l, int& d, int& h, int& w) { d = dim == 3 ? D >> l : 1; h = H >> l; w = W >> l; };
    auto vox3 = [&](int l) { return (long long)(dim == 3 ? D >> l : 1) * (H >> l) * (W >> l); };
    auto convm = [&](const ConvOp& op, long long xo, long long x_ss, long long x8o, long long x8_ss, long long yo, long long y_ss, int y_lo,
                     long long y8o, long long y8_ss, int l) -> int {
      int d, h, w;
      dims3(l, d, h, w);
      const float* aux = (const float*)(K + op.aux);
      return iunet_x2m_conv_fwd(dim, WS + xo, x_ss, WS + x8o, x8_ss, WS + yo, y_ss, y_lo, y8o >= 0 ? WS + y8o : nullptr, y8_ss, K + op.pk[1], K + op.pk[0],
                                aux, aux + op.co, N, d, h, w, op.ci, op.co, 2, sat, stream);
    };
    int rc = 0;
    for (int l = 0; l < lv; ++l) {
      int d, h, w;
      dims3(l, d, h, w);
      const long long v = vox3(l);
      const int c = n->ch[l];
      const ConvOp& c1 = n->conv[2 * stage_index(n, false, l)];
      const ConvOp& c2 = n->conv[2 * stage_index(n, false, l) + 1];
      if (l == 0 && lv > 1 && iunet_x2m_first_stage_fusable(dim, n->cin, c, N, h, w)) {
        // 2-D, one input channel: the first encoder stage is ONE launch (the first conv is computed by the second conv's loader waves)
        const float* aux1 = (const float*)(K + c1.aux);
        const float* aux2 = (const float*)(K + c2.aux);
        const bool pooled = iunet_x2m_pool_fusable(dim, c) != 0;
        rc = iunet_x2m_first_stage_fwd(x, in_dtype, in_strides, K + c1.pk[1], aux1, aux1 + c, A, WS + L.cat[0], 2ll * c * v, -1, WS + L.catm[0], 2ll * c * v,
                                       pooled ? WS + L.pin[1] : nullptr, (long long)c * vox3(1), pooled ? WS + L.pinm[1] : nullptr, (long long)c * vox3(1),
                                       K + c2.pk[1], K + c2.pk[0], aux2, aux2 + c2.co, N, h, w, sat, stream);
        if (rc) return rc;
        if (!pooled) {
          int dn, hn, wn;
          dims3(1, dn, hn, wn);
          rc = iunet_x2m_maxpool_fwd(dim, WS + L.cat[0], 2ll * c * v, WS + L.catm[0], 2ll * c * v, WS + L.pin[1], (long long)c * vox3(1), WS + L.pinm[1],
                                     (long long)c * vox3(1), c, N, dn, hn, wn, stream);
          if (rc) return rc;
        }
        continue;
      }
      if (l == 0) {
        const float* aux = (const float*)(K + c1.aux);
        rc = iunet_x2m_first_conv_fwd(dim, x, in_dtype, in_strides, WS + L.a[0], (long long)c * v, -1, WS + L.am[0], (long long)c * v, K + c1.pk[1], aux,
                                      aux + c, A, N, d, h, w, n->cin, c, 1, sat, stream);
      } else {
        const int cp = n->ch[l - 1];
        rc = convm(c1, L.pin[l], (long long)cp * v, L.pinm[l], (long long)cp * v, L.a[l], (long long)c * v, -1, L.am[l], (long long)c * v, l);
      }
      if (rc) return rc;
      if (l < lv - 1) {
        if (iunet_x2m_pool_fusable(dim, c)) {       // the stage's max-pool rides in the epilogue of its second conv (same words, one launch)
          const float* aux = (const float*)(K + c2.aux);
          rc = iunet_x2m_conv_pool_fwd(dim, WS + L.a[l], (long long)c * v, WS + L.am[l], (long long)c * v, WS + L.cat[l], 2ll * c * v, -1, WS + L.catm[l],
                                       2ll * c * v, WS + L.pin[l + 1], (long long)c * vox3(l + 1), WS + L.pinm[l + 1], (long long)c * vox3(l + 1),
                                       K + c2.pk[1], K + c2.pk[0], aux, aux + c2.co, N, d, h, w, c2.ci, c2.co, 2, sat, stream);
        } else {
          rc = convm(c2, L.a[l], (long long)c * v, L.am[l], (long long)c * v, L.cat[l], 2ll * c * v, -1, L.catm[l], 2ll * c * v, l);
          if (rc) return rc;
          int dn, hn, wn;
          dims3(l + 1, dn, hn, wn);
          rc = iunet_x2m_maxpool_fwd(dim, WS + L.cat[l], 2ll * c * v, WS + L.catm[l], 2ll * c * v, WS + L.pin[l + 1], (long long)c * vox3(l + 1),
                                     WS + L.pinm[l + 1], (long long)c * vox3(l + 1), c, N, dn, hn, wn, stream);
        }
      } else {
        rc = convm(c2, L.a[l], (long long)c * v, L.am[l], (long long)c * v, L.b[l], 2ll * c * v, c / 8, -1, 0, l);
      }
      if (rc) return rc;
    }
    for (int l = lv - 2; l >= 0; --l) {
      int di, hi, wi;
      dims3(l + 1, di, hi, wi);
      const long long v = vox3(l), vi = vox3(l + 1);
      const int c = n->ch[l], cn = n->ch[l + 1];
      const UpOp& u = n->up[lv - 2 - l];
      const float* aux = (const float*)(K + u.aux);
      // up half of the concat buffer: hi planes [c / 8, 2 c / 8), lo8 planes [c / 16, 2 c / 16)
      rc = iunet_x2m_convT_fwd(dim, WS + L.b[l + 1], 2ll * cn * vi, cn / 8, WS + L.cat[l] + (long long)(c / 8) * v * 16, 2ll * c * v, -1,
                               WS + L.catm[l] + (long long)(c / 16) * v * 16, 2ll * c * v, K + u.pk, aux, aux + c, N, di, hi, wi, cn, c, sat, stream);
      if (rc) return rc;
      const ConvOp& c1 = n->conv[2 * stage_index(n, true, l)];
      const ConvOp& c2 = n->conv[2 * stage_index(n, true, l) + 1];
      rc = convm(c1, L.cat[l], 2ll * c * v, L.catm[l], 2ll * c * v, L.a[l], (long long)c * v, -1, L.am[l], (long long)c * v, l);
      if (rc) return rc;
      if (l == 0 && (logits || probs || cls) && iunet_x2m_head_fusable(n->ncls, c)) {
        // the head in the last conv's epilogue: the last activation is never written (the same bits as conv + head)
        const long long dflt0[5] = {n->ncls * v, v, (long long)H * W, W, 1};
        const float* aux2 = (const float*)(K + c2.aux);
        return iunet_x2m_conv_head_fwd(dim, WS + L.a[0], (long long)c * v, WS + L.am[0], (long long)c * v, K + c2.pk[1], K + c2.pk[0], aux2, aux2 + c,
                                       n->flat + n->head_w, n->flat + n->head_b, A, n->ncls, logits, probs, cls, out_strides ? out_strides : dflt0,
                                       divisor, accumulate, N, dim == 3 ? D : 1, H, W, c, sat, stream);
      }
      rc = convm(c2, L.a[l], (long long)c * v, L.am[l], (long long)c * v, L.b[l], 2ll * c * v, c / 8, -1, 0, l);
      if (rc) return rc;
    }
    if (!logits && !probs && !cls) return IUNET_OK;
    const long long v0 = vox3(0);
    const long long dflt[5] = {n->ncls * v0, v0, (long long)H * W, W, 1};
    const int c0 = n->ch[0];
    return iunet_x2_head_fwd(WS + L.b[0], 2ll * c0 * v0, c0 / 8, c0, n->flat + n->head_w, n->flat + n->head_b, A, n->ncls, logits, probs, cls,
                             out_strides ? out_strides : dflt, divisor, accumulate, N, D, H, W, stream);
  }
  const bool x2 = mode == 2;
  const int mul = x2 ? 2 : 1;
  auto dims = [&](int l, int& d, int& h, int& w) { d = dim == 3 ? D >> l : 1; h = H >> l; w = W >> l; };
  auto vox = [&](int l) { int d, h, w; dims(l, d, h, w); return (long long)d * h * w; };
  auto plane = [&](long long base_off, int planes, int l) { return (void*)(WS + base_off + (long long)planes * vox(l) * 16); };
  int rc = 0;
  // one stage conv: x view (ptr, sample stride, lo-plane offset) -> y view
  auto conv = [&](const ConvOp& op, const void* xp, long long x_ss, int x_lo, void* yp, long long y_ss, int y_lo, int l) -> int {
    int d, h, w;
    dims(l, d, h, w);
    const float* aux = (const float*)(K + op.aux);
    if (x2 && n->norm == 1) {        // raw output (no bias, no ReLU) as split words, then statistics + normalise + ReLU into the consumer's view
      const long long v = (long long)d * h * w;
      int r = iunet_x2_conv3_fwd_flag(dim, xp, x_ss, x_lo, WS + L.raw, 2ll * op.co * v, op.co / 8, K + op.pk[1], aux, aux + op.co, N, d, h, w, op.ci, op.co, 0, WS, stream);
      if (r) return r;
      return iunet_x2_gn_relu_fwd(WS + L.raw, 2ll * op.co * v, op.co / 8, yp, y_ss, y_lo, n->flat + op.bn, n->flat + op.bn + op.co, n->groups, 1e-5f,
                                  n->act_scale, WS + L.gnslab, WS + L.gnsc, WS + L.gnsh, op.co, N, v, WS, stream);
    }
    if (x2) return iunet_x2_conv3_fwd_flag(dim, xp, x_ss, x_lo, yp, y_ss, y_lo, K + op.pk[1], aux, aux + op.co, N, d, h, w, op.ci, op.co, 2, WS, stream);
    int lay = 1;
    if (op.pk[3] >= 0 && iunet_conv3_compact_ok(dim, N, d, h, w, op.ci, op.co, 0, 0)) lay = 3;
    else {
      lay = iunet_conv3_pick_layout(dim, N, d, h, w, op.ci, op.co);
      if (lay == 0 && op.pk[0] < 0) lay = 1;
    }
    return iunet_conv3_fwd(mode, dim, xp, x_ss, yp, y_ss, K + op.pk[lay == 2 ? 1 : lay], aux + op.co, nullptr, N, d, h, w, op.ci, op.co, 2, lay, stream);
  };
  for (int l = 0; l < lv; ++l) {
    int d, h, w;
    dims(l, d, h, w);
    const long long v = vox(l);
    const int c = n->ch[l], c8 = c / 8;
    const ConvOp& c1 = n->conv[2 * stage_index(n, false, l)];
    const ConvOp& c2 = n->conv[2 * stage_index(n, false, l) + 1];
    if (l == 0) {
      const float* aux = (const float*)(K + c1.aux);
      const bool gn = n->norm == 1;
      rc = x2 ? iunet_x2m_first_conv_fwd(dim, x, in_dtype, in_strides, WS + (gn ? L.raw : L.a[0]), 2ll * c * v, c8, nullptr, 0, K + c1.pk[1], aux, aux + c, n->act_scale,
                                         N, d, h, w, n->cin, c, gn ? 0 : 1, WS, stream)
              : iunet_first_conv_fwd(mode, dim, x, in_dtype, in_strides, WS + L.a[0], (long long)c * v, K + c1.pk[1], aux + c, nullptr,
                                     N, d, h, w, n->cin, c, 1, stream);
      if (!rc && x2 && gn)
        rc = iunet_x2_gn_relu_fwd(WS + L.raw, 2ll * c * v, c8, WS + L.a[0], 2ll * c * v, c8, n->flat + c1.bn, n->flat + c1.bn + c, n->groups, 1e-5f, n->act_scale,
                                  WS + L.gnslab, WS + L.gnsc, WS + L.gnsh, c, N, v, WS, stream);
    } else {
      const int cp = n->ch[l - 1];
      rc = conv(c1, WS + L.pin[l], (long long)mul * cp * v, cp / 8, WS + L.a[l], (long long)mul * c * v, c8, l);
    }
    if (rc) return rc;
    if (l < lv - 1) {
      // skip half of the concat buffer: hi planes [0, c8) (x2: lo planes [2 c8, 3 c8))
      rc = conv(c2, WS + L.a[l], (long long)mul * c * v, c8, WS + L.cat[l], (long long)mul * 2 * c * v, 2 * c8, l);
      if (rc) return rc;
      int dn, hn, wn;
      dims(l + 1, dn, hn, wn);
      rc = x2 ? iunet_x2_maxpool_fwd(dim, WS + L.cat[l], 4ll * c * v, 2 * c8, WS + L.pin[l + 1], 2ll * c * vox(l + 1), c8, c, N, dn, hn, wn, stream)
              : iunet_maxpool_fwd(mode, dim, WS + L.cat[l], 2ll * c * v, WS + L.pin[l + 1], (long long)c * vox(l + 1), c, N, dn, hn, wn, stream);
    } else {
      rc = conv(c2, WS + L.a[l], (long long)mul * c * v, c8, WS + L.b[l], (long long)mul * c * v, c8, l);
    }
    if (rc) return rc;
  }
  for (int l = lv - 2; l >= 0; --l) {
    int d, h, w, di, hi, wi;
    dims(l, d, h, w);
    dims(l + 1, di, hi, wi);
    const long long v = vox(l), vi = vox(l + 1);
    const int c = n->ch[l], c8 = c / 8, cn = n->ch[l + 1];
    const UpOp& u = n->up[lv - 2 - l];
    const float* aux = (const float*)(K + u.aux);
    // up half of the concat buffer: hi planes [c8, 2 c8) (x2: lo planes [3 c8, 4 c8))
    rc = x2 ? iunet_x2m_convT_fwd(dim, WS + L.b[l + 1], 2ll * cn * vi, cn / 8, plane(L.cat[l], c8, l), 4ll * c * v, 2 * c8, nullptr, 0, K + u.pk, aux, aux + c,
                                  N, di, hi, wi, cn, c, WS, stream)
            : iunet_convT_fwd(mode, dim, WS + L.b[l + 1], (long long)cn * vi, plane(L.cat[l], c8, l), 2ll * c * v, K + u.pk, n->flat + u.b,
                              N, di, hi, wi, cn, c, stream);
    if (rc) return rc;
    const ConvOp& c1 = n->conv[2 * stage_index(n, true, l)];
    const ConvOp& c2 = n->conv[2 * stage_index(n, true, l) + 1];
    rc = conv(c1, WS + L.cat[l], (long long)mul * 2 * c * v, 2 * c8, WS + L.a[l], (long long)mul * c * v, c8, l);
    if (rc) return rc;
    rc = conv(c2, WS + L.a[l], (long long)mul * c * v, c8, WS + L.b[l], (long long)mul * c * v, c8, l);
    if (rc) return rc;
  }
  if (!logits && !probs && !cls) return IUNET_OK;
  const long long v0 = vox(0);
  const long long dflt[5] = {n->ncls * v0, v0, (long long)H * W, W, 1};
  const long long* os = out_strides ? out_strides : dflt;
  const int c0 = n->ch[0];
  return x2 ? iunet_x2_head_fwd(WS + L.b[0], 2ll * c0 * v0, c0 / 8, c0, n->flat + n->head_w, n->flat + n->head_b, n->act_scale, n->ncls, logits,
                                probs, cls, os, divisor, accumulate, N, D, H, W, stream)
            : iunet_head_fwd(mode, WS + L.b[0], (long long)c0 * v0, c0, n->flat + n->head_w, n->flat + n->head_b, n->ncls, logits, probs, cls, os,
                             divisor, accumulate, N, D, H, W, stream);
}

/* validation_step (unet.py:104-116) in one call, 16-bit modes (0 / 1): the eval-mode forward (BatchNorm running statistics folded into the
 * operators by iunet_net_load) up to the head's input, then the fused head + softmax + loss of iunet_head_loss_fwd on target / weight
 * [N][ncls][D*H*W] (tdtype 0 f32, 1 f16; weight may be null; kind: metrics.py's seven losses) -> out4 = [Loss, Dice, IoU, MCC] (fp32,
 * device).  scratch: iunet_net_eval_scratch_bytes. */
long long iunet_net_eval_scratch_bytes(const iunet_net* n, int N, int D, int H, int W) {
  if (!n || iunet_net_workspace_bytes(n, N, D, H, W) <= 0) return 0;
  const long long v0 = (long long)D * H * W;
  return align256((long long)iunet_head_loss_num_parts(N, v0) * n->ncls * 8 * 4) + align256((long long)n->ncls * 3 * 4);
}
int iunet_net_eval_step(iunet_net* n, const void* x, int in_dtype, const long long* in_strides, const void* target, const void* weight, int tdtype,
                        int loss_kind, int N, int D, int H, int W, void* workspace, void* scratch, void* out4, void* stream) {
  IUNET_REQUIRE(n && target && scratch && out4, "net_eval_step: null pointer");
  IUNET_REQUIRE(n->mode == 0 || n->mode == 1, "net_eval_step: the validation step runs in the 16-bit training dtype (mode 0 / 1, got %d)", n->mode);
  IUNET_REQUIRE(tdtype == 0 || tdtype == 1, "net_eval_step: target dtype must be 0 (f32) or 1 (f16)");
  IUNET_REQUIRE(loss_kind >= 0 && loss_kind <= 6, "net_eval_step: loss kind %d", loss_kind);
  int rc = iunet_net_forward(n, x, in_dtype, in_strides, N, D, H, W, workspace, nullptr, nullptr, nullptr, nullptr, 1.0f, 0, stream);
  if (rc) return rc;
  const WsLayout L = ws_layout(n, N, D, H, W);
  const long long v0 = (long long)D * H * W;
  const int c0 = n->ch[0];
  unsigned char* S = (unsigned char*)scratch;
  void* coef = S + align256((long long)iunet_head_loss_num_parts(N, v0) * n->ncls * 8 * 4);
  return iunet_head_loss_fwd(n->mode, (unsigned char*)workspace + L.b[0], (long long)c0 * v0, c0, n->flat + n->head_w, n->flat + n->head_b, n->ncls,
                             target, weight, tdtype, loss_kind, S, out4, coef, N, v0, stream);
}

/* predict.py:30-38 in one call: uint8 [N][cin][D][H][W] (contiguous) -> class map uint8 [N][D*H*W] */
int iunet_net_forward_argmax(iunet_net* n, const void* x_u8, void* cls_u8, int N, int D, int H, int W, void* workspace, void* stream) {
  IUNET_REQUIRE(n && x_u8 && cls_u8, "net_forward_argmax: null pointer");
  const long long v = (long long)D * H * W;
  const long long st[5] = {n->cin * v, v, (long long)H * W, W, 1};
  return iunet_net_forward(n, x_u8, 2, st, N, D, H, W, workspace, nullptr, nullptr, cls_u8, nullptr, 1.0f, 0, stream);
}

}  // extern "C"
