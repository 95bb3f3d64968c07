// Handle-level C ABI of the TRAINING step: unet.py:88-102 (training_step: forward with BatchNorm batch statistics, the reference's loss
// on softmax probabilities) + the backward pass + AdamW (unet.py:71-73) + the re-pack of the updated operators, sequenced in C++ -- what
// interactive_unet/train_engine.py sequences from Python (~130 launches per step), from the same entry points in the same order, so the
// two are bit-identical (tests/test_train_handle.py).  For a caller that is not Python, and for the small 2-D steps of the reference's
// UI loop (app.py:203-210: batch 8 of 512^2, trainer.py:56-63), whose launches are shorter than the Python between them.
//
// Everything on the device is the caller's: the flat fp32 parameter / gradient / moment vectors (canonical order, iunet_train_param),
// the BatchNorm running statistics (one pointer pair per BatchNorm), the packed operators, the workspace of one step and the 32-byte
// training state (loss scale, step count, overflow back-off: train_pointwise.hip) -- the fp16 loss scale and the skipped step on
// overflow are handled on the device, nothing is read back per step.  The handle itself is host memory; no allocation, no
// synchronisation in a step.  BatchNorm networks (the GroupNorm variant stays on the Python-sequenced path).
#include "common.h"
#include "pack_desc.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

extern "C" {
int iunet_pack_batch(const void*, int, int, void*);
int iunet_pack_desc_bytes(void);
long long iunet_pack_conv3_elems(int, int, int, int);
long long iunet_pack_first_conv_elems(int, int, int);
int iunet_conv3_pick_layout(int, int, int, int, int, int, int);
int iunet_conv3_compact_ok(int, int, int, int, int, int, int, int, int);
int iunet_conv3_num_tiles(int, int, int, int, int);
int iunet_conv3_stats_parts(int, int, int, int, int, int, int);
int iunet_first_conv_fwd(int, int, const void*, int, const long long*, void*, long long, const void*, const void*, void*, int, int, int,
                         int, int, int, int, void*);
int iunet_conv3_fwd(int, int, const void*, long long, void*, long long, const void*, const void*, void*, int, int, int, int, int, int,
                    int, int, void*);
int iunet_conv3_fwd_act(int, int, const void*, long long, void*, long long, const void*, const void*, void*, const void*, const void*,
                        int, int, int, int, int, int, int, int, void*);
int iunet_convT_fwd(int, int, const void*, long long, void*, long long, const void*, const void*, int, int, int, int, int, int, void*);
int iunet_bn_finalize(const void*, int, int, double, const void*, const void*, void*, void*, float, float, void*, void*, void*, void*, void*);
int iunet_bn_relu_fwd(int, const void*, long long, void*, long long, const void*, const void*, int, int, long long, void*);
int iunet_bn_relu_pool_fwd(int, int, const void*, long long, void*, long long, void*, long long, const void*, const void*, int, int, int,
                           int, int, void*);
int iunet_bn_bwd_num_parts(int, long long);
int iunet_bn_relu_bwd(int, const void*, long long, const void*, long long, const void*, long long, void*, long long, const void*,
                      const void*, const void*, const void*, const void*, void*, void*, void*, void*, int, int, long long, void*);
int iunet_conv3_dgrad_bnstats_lay(int, int, const void*, long long, void*, long long, const void*, void*, const void*, long long, const void*,
                                  const void*, const void*, const void*, int, int, int, int, int, int, int, void*);
int iunet_bn_relu_bwd_apply(int, const void*, long long, const void*, long long, void*, long long, const void*, const void*, const void*,
                            const void*, const void*, void*, void*, const void*, int, void*, int, int, long long, void*);
int iunet_bn_relu_pool_bwd(int, int, const void*, long long, const void*, long long, const void*, long long, void*, long long, const void*,
                           const void*, const void*, const void*, const void*, void*, void*, void*, void*, int, int, int, int, int, void*);
int iunet_head_loss_num_parts(int, long long);
int iunet_head_loss_fwd(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, int, void*, void*,
                        void*, int, long long, void*);
int iunet_head_loss_fwd_act(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, int, void*,
                            void*, void*, const void*, const void*, int, long long, void*);
int iunet_head_loss_bwd_num_parts(int, long long, int, int);
int iunet_head_loss_bwd_dev(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, const void*,
                            const void*, void*, long long, void*, const void*, const void*, int, long long, void*);
int iunet_head_grad_scatter(const void*, void*, void*, int, int, void*);
int iunet_head_bn_bwd_ok(int, int);
int iunet_head_gn_bwd(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, const void*, float, const void*, const void*,
                      const void*, const void*, const void*, const void*, int, void*, void*, void*, long long, void*, void*, void*, void*, int, long long, void*);
int iunet_head_loss_fwd_act_ps(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, int, void*, void*, void*,
                               const void*, const void*, int, int, long long, void*);
int iunet_head_bn_bwd(int, const void*, long long, int, const void*, const void*, int, const void*, const void*, int, const void*, float, const void*, const void*,
                      const void*, const void*, const void*, const void*, void*, void*, void*, long long, void*, void*, void*, void*, int, long long, void*);
int iunet_reduce_slab(void*, int, long long, void*, float, int, void*);
long long iunet_conv3_wgrad_slab_floats(int, int, int, int, int, int, int);
int iunet_conv3_wgrad(int, int, const void*, long long, const void*, long long, void*, void*, float, int, int, int, int, int, int, void*);
int iunet_conv3_wgrad_act(int, int, const void*, long long, const void*, long long, void*, void*, float, const void*, const void*, int,
                          int, int, int, int, int, void*);
int iunet_convT_dgrad(int, int, const void*, long long, void*, long long, const void*, int, int, int, int, int, int, void*);
int iunet_convT_wgrad_blocks(int, int, int, int, int, int, int);
int iunet_convT_wgrad(int, int, const void*, long long, const void*, long long, void*, void*, void*, void*, int, int, int, int, int, int,
                      void*);
int iunet_first_conv_wgrad_blocks(int, int, int, int, int);
int iunet_first_conv_wgrad_bn(int, int, const void*, int, const long long*, const void*, long long, const void*, long long, const void*,
                              const void*, const void*, const void*, const void*, void*, void*, int, int, int, int, int, int, void*);
int iunet_adamw_step_dev(void*, const void*, void*, void*, long long, float, float, float, float, float, void*, int, float, void*);
int iunet_conv3_sample_stats_rows(int, int, int, int, int, int, int, int, int);
int iunet_conv3_fwd_sample_stats(int, int, const void*, long long, void*, long long, const void*, void*, int, int, int, int, int, int, int, void*);
int iunet_conv3_dgrad_sample_bnstats(int, int, const void*, long long, void*, long long, const void*, void*, const void*, long long, const void*, const void*,
                                     const void*, const void*, int, int, int, int, int, int, int, void*);
int iunet_gn_relu_bwd_rows(int, const void*, long long, const void*, long long, void*, long long, const void*, int, const void*, const void*, const void*,
                           const void*, void*, void*, void*, int, void*, int, int, long long, void*);
int iunet_gn_relu_fwd_rows(int, const void*, long long, void*, long long, const void*, const void*, int, float, void*, int, void*, void*, void*, void*, int, int,
                           long long, void*);
int iunet_gn_relu_pool_fwd_rows(int, int, const void*, long long, void*, long long, void*, long long, const void*, const void*, int, float, void*, int, void*,
                                void*, void*, void*, int, int, int, int, int, void*);
int iunet_gn_relu_fwd(int, const void*, long long, void*, long long, const void*, const void*, int, float, void*, void*, void*, void*, void*, int, int,
                      long long, void*);
int iunet_gn_relu_pool_fwd(int, int, const void*, long long, void*, long long, void*, long long, const void*, const void*, int, float, void*, void*,
                           void*, void*, void*, int, int, int, int, int, void*);
int iunet_gn_relu_bwd(int, const void*, long long, const void*, long long, void*, long long, const void*, int, const void*, const void*, const void*,
                      const void*, void*, void*, void*, void*, int, int, long long, void*);
int iunet_gn_relu_pool_bwd(int, int, const void*, long long, const void*, long long, const void*, long long, void*, long long, const void*, int,
                           const void*, const void*, const void*, const void*, void*, void*, void*, void*, int, int, int, int, int, void*);
int iunet_first_conv_wgrad(int, int, const void*, int, const long long*, const void*, long long, void*, void*, int, int, int, int, int, int, void*);
}

namespace {

long long align256(long long v) { return (v + 255) & ~255ll; }
bool env_on(const char* name) { const char* e = getenv(name); return e != nullptr && e[0] != 0; }

struct TParam { std::string name; long long off, numel; };

// one operator in the fragment order(s) its launches may need (interactive_unet/_native.py: PackedConv)
struct TPack {
  int cout = 0, cin = 0, dg = 0;
  long long buf[4] = {-1, -1, -1, -1};          // packed-buffer byte offsets of layouts 0, 1 (K16, also layout 2) and 3 (compact)
  long long elems[4] = {0, 0, 0, 0};
};

struct TConv {
  std::string name;
  int ci, co, l, first;
  long long w, gamma, beta;                      // flat offsets
  int bn;                                        // index of its BatchNorm (running statistics pointer pair)
  TPack fwd, dgr;
  long long first_pk = -1, first_elems = 0;      // the first conv's operator
};
struct TUp { int ci, co, l; long long w, b, fwd, dgr; };

struct TWs {
  std::vector<long long> y, z, dz, scale, shift, mean, invstd;      // per conv (z / dz: -1 for the skip convs)
  std::vector<long long> cat, dcat, bslab, pin, dpin;               // per level
  long long dy, stats, wslab, bnslab, bncoef, lslab, hslab, htmp, out4, coef, bytes;
};

}  // namespace

struct iunet_train {
  int norm = 0, groups = 8;                      // norm 1: GroupNorm(groups) after every stage conv (statistics per (sample, group), nothing fused into the convs)
  int dim, levels, base, cin, ncls, dtype, kind;
  int taps, npos;
  bool fuse_act, fuse_bw, head_act, gn_conv_stats, gn_bw, head_bn, gn_head;
  std::vector<int> ch;
  std::vector<TParam> params;
  long long nparams = 0;
  std::vector<TConv> conv;                       // enc0.conv1, enc0.conv2, ..., dec{L-2}.conv1, ... (stage order)
  std::vector<TUp> up;                           // dec{L-2}.up ... dec0.up
  long long head_w = 0, head_b = 0;
  long long packed_bytes = 0, table_off = 0;
  int ndesc = 0;
  // bound device buffers
  float *flat = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr, *state = nullptr;
  std::vector<float*> running;                   // [2 * BatchNorms]: mean, var
  unsigned char* packed = nullptr;
};

namespace {

int stage_index(const iunet_train* n, bool dec, int l) { return dec ? n->levels + (n->levels - 2 - l) : l; }
const TConv& conv_of(const iunet_train* n, bool dec, int l, int j) { return n->conv[2 * stage_index(n, dec, l) + (j - 1)]; }

void pack_alloc(TPack& p, int cout, int cin, int taps, int dg, long long& pk) {
  p.cout = cout; p.cin = cin; p.dg = dg;
  const int out_ch = dg ? cin : cout, in_ch = dg ? cout : cin;
  auto take = [&](int lay, int mode) { p.elems[lay] = iunet_pack_conv3_elems(cout, cin, taps, mode | dg); p.buf[lay] = pk; pk = align256(pk + p.elems[lay] * 2); };
  take(1, 2);
  const bool compact2d = taps == 9 && !env_on("IUNET_NO_COMPACT2D");
  if (out_ch % 64 == 0 && taps == 9 && in_ch > 64 && !compact2d) take(0, 0);
  if (((taps == 27 && in_ch > 32) || compact2d) && !env_on("IUNET_NO_COMPACT")) take(3, 6);
}

// (layout, packed-buffer offset) of a launch on this grid (PackedConv.pick)
int pack_pick(const TPack& p, int nd, int N, int D, int H, int W, bool act, bool bw, long long* off) {
  const int in_ch = p.dg ? p.cout : p.cin, out_ch = p.dg ? p.cin : p.cout;
  int lay = iunet_conv3_pick_layout(nd, N, D, H, W, in_ch, out_ch);
  bw = bw && lay == 2;
  if (p.buf[3] >= 0 && iunet_conv3_compact_ok(nd, N, D, H, W, in_ch, out_ch, act ? 1 : 0, bw ? 1 : 0)) { *off = p.buf[3]; return 3; }
  if (lay == 0 && p.buf[0] < 0) lay = 1;
  *off = p.buf[lay == 2 ? 1 : lay];
  return lay;
}

TWs ws_layout(const iunet_train* n, int N, int D, int H, int W) {
  TWs L;
  const int lv = n->levels, dim = n->dim;
  long long off = 0;
  auto act = [&](long long elems) { const long long o = off; off = align256(off + elems * 2); return o; };
  auto f32 = [&](long long nfl) { const long long o = off; off = align256(off + nfl * 4); return o; };
  auto dims = [&](int l, int& d, int& h, int& w) { d = dim == 3 ? D >> l : 1; h = H >> l; w = W >> l; };
  auto vox = [&](int l) { int d, h, w; dims(l, d, h, w); return (long long)d * h * w; };
  const size_t nc = n->conv.size();
  L.y.assign(nc, -1); L.z.assign(nc, -1); L.dz.assign(nc, -1);
  L.scale.assign(nc, -1); L.shift.assign(nc, -1); L.mean.assign(nc, -1); L.invstd.assign(nc, -1);
  long long max_stats = 0, max_wslab = 0, max_bn = 0, max_dy = 0;
  for (size_t k = 0; k < nc; ++k) {
    const TConv& c = n->conv[k];
    int d, h, w;
    dims(c.l, d, h, w);
    const long long v = vox(c.l);
    L.y[k] = act((long long)N * c.co * v);
    const bool enc = c.name[0] == 'e', skip = enc && c.name.back() == '2' && c.l < lv - 1;
    if (!skip) { L.z[k] = act((long long)N * c.co * v); L.dz[k] = act((long long)N * c.co * v); }
    const long long rows = n->norm == 1 ? N : 1;          // GroupNorm: one (scale, shift, mean, invstd) row per sample
    L.scale[k] = f32(rows * c.co); L.shift[k] = f32(rows * c.co); L.mean[k] = f32(rows * c.co); L.invstd[k] = f32(rows * c.co);
    if (c.first) {
      max_stats = std::max(max_stats, (long long)iunet_conv3_num_tiles(dim, N, d, h, w) * c.co * 2);
      max_wslab = std::max(max_wslab, (long long)iunet_first_conv_wgrad_blocks(dim, N, d, h, w) * c.co * 112);
    } else {
      const long long p0 = iunet_conv3_stats_parts(dim, N, d, h, w, c.co, 0), p2 = iunet_conv3_stats_parts(dim, N, d, h, w, c.co, 2);
      max_stats = std::max(max_stats, std::max(p0, p2) * c.co * 2);
      if (n->norm == 1)       // per-sample rows of the conv epilogue (layouts 2 and 3 share the grid)
        for (int lay = 2; lay <= 3; ++lay)
          max_stats = std::max(max_stats, (long long)N * iunet_conv3_sample_stats_rows(n->dtype, dim, N, d, h, w, c.ci, c.co, lay) * c.co * 2);
      max_wslab = std::max(max_wslab, iunet_conv3_wgrad_slab_floats(dim, N, d, h, w, c.ci, c.co));
    }
    max_bn = std::max(max_bn, (long long)iunet_bn_bwd_num_parts(N, v) * c.co * 2);
    max_dy = std::max(max_dy, (long long)n->ch[c.l] * v);
  }
  L.cat.assign(lv, -1); L.dcat.assign(lv, -1); L.bslab.assign(lv, -1); L.pin.assign(lv, -1); L.dpin.assign(lv, -1);
  for (int l = 0; l < lv; ++l) {
    const long long v = vox(l);
    if (l < lv - 1) {
      L.cat[l] = act((long long)N * 2 * n->ch[l] * v);
      L.dcat[l] = act((long long)N * 2 * n->ch[l] * v);
      int d, h, w;
      dims(l + 1, d, h, w);
      const long long nb = iunet_convT_wgrad_blocks(dim, N, d, h, w, n->ch[l + 1], n->ch[l]);
      max_wslab = std::max(max_wslab, nb * n->ch[l + 1] * n->ch[l] * n->npos);
      L.bslab[l] = f32(nb * n->ch[l]);
    }
    if (l > 0) { L.pin[l] = act((long long)N * n->ch[l - 1] * v); L.dpin[l] = act((long long)N * n->ch[l - 1] * v); }
  }
  const long long v0 = vox(0);
  L.dy = act((long long)N * max_dy);
  L.stats = f32(max_stats); L.wslab = f32(max_wslab); L.bnslab = f32(max_bn);
  L.bncoef = f32(3ll * n->ch[lv - 1] * (n->norm == 1 ? N : 1));
  L.lslab = f32((long long)iunet_head_loss_num_parts(N, v0) * n->ncls * 8);
  L.hslab = f32((long long)iunet_head_loss_bwd_num_parts(N, v0, n->ncls, n->ch[0]) * n->ncls * (n->ch[0] + 1));
  L.htmp = f32((long long)n->ncls * (n->ch[0] + 1));
  L.out4 = f32(4);
  L.coef = f32((long long)n->ncls * 3);
  L.bytes = off;
  return L;
}

}  // namespace

extern "C" {

/* dtype: 0 fp16, 1 bf16 (the 16-bit training modes of interactive_unet.train_engine.TrainEngine); loss_kind: 0 ce, 1 dice, 2 iou, 3 mcc,
 * 4 dice_ce, 5 iou_ce, 6 mcc_ce (utils.py:458-475; the reference's default is mcc_ce, unet.py:17) */
int iunet_train_create_ex(int dim, int levels, int base, int cin, int ncls, int dtype, int loss_kind, int norm, int groups, iunet_train** out);
int iunet_train_create(int dim, int levels, int base, int cin, int ncls, int dtype, int loss_kind, iunet_train** out) {
  return iunet_train_create_ex(dim, levels, base, cin, ncls, dtype, loss_kind, 0, 8, out);
}
/* norm: 0 BatchNorm, 1 GroupNorm(groups) after every stage conv (north star "GroupNorm/BN"; statistics per (sample, group), the same at
 * training and inference -- the running-statistics pointers of iunet_train_bind are accepted and left alone) */
int iunet_train_create_ex(int dim, int levels, int base, int cin, int ncls, int dtype, int loss_kind, int norm, int groups, iunet_train** out) {
  IUNET_REQUIRE(out != nullptr, "train_create: null handle pointer");
  IUNET_REQUIRE(norm == 0 || (norm == 1 && groups > 0 && base % groups == 0), "train_create: norm must be 0 (batch) or 1 (group, groups dividing base): %d, %d groups", norm, groups);
  IUNET_REQUIRE(dim == 2 || dim == 3, "train_create: dim must be 2 or 3 (got %d)", dim);
  IUNET_REQUIRE(levels >= 2 && levels <= 6, "train_create: levels must be 2..6 (got %d)", levels);
  IUNET_REQUIRE(base > 0 && base % 32 == 0, "train_create: base channels must be a positive multiple of 32 (got %d)", base);
  IUNET_REQUIRE(base == 32 || base == 64, "train_create: the fused head + loss kernels take 32 or 64 head input channels (base %d)", base);
  IUNET_REQUIRE(cin >= 1 && cin <= 4, "train_create: 1..4 input channels (got %d)", cin);
  IUNET_REQUIRE(ncls >= 2 && ncls <= 10, "train_create: 2..10 classes (app.py:162; got %d)", ncls);
  IUNET_REQUIRE(dtype == 0 || dtype == 1, "train_create: dtype must be 0 (fp16) or 1 (bf16), got %d", dtype);
  IUNET_REQUIRE(loss_kind >= 0 && loss_kind <= 6, "train_create: loss kind must be 0..6 (got %d)", loss_kind);
  IUNET_REQUIRE(iunet_pack_desc_bytes() == (int)sizeof(PackDesc), "train_create: PackDesc layout mismatch");
  iunet_train* n = new iunet_train();
  n->dim = dim; n->levels = levels; n->base = base; n->cin = cin; n->ncls = ncls; n->dtype = dtype; n->kind = loss_kind;
  n->taps = dim == 3 ? 27 : 9; n->npos = dim == 3 ? 8 : 4;
  // the A/B switches of the Python-sequenced engine (train_engine.TrainEngine.__init__): the two sequences stay the same launches
  n->norm = norm; n->groups = groups;
  // (GroupNorm: per-sample statistics -- none of the per-channel fusions of the BatchNorm path applies, train_engine.TrainEngine.__init__)
  n->fuse_act = norm == 0 && !env_on("IUNET_NO_ACT_FUSION");
  n->fuse_bw = norm == 0 && !env_on("IUNET_NO_BW_FUSION");
  n->head_act = norm == 0 && !env_on("IUNET_NO_HEAD_ACT");
  // GroupNorm: the head reads the last conv's raw output with per-sample rows and its backward runs as iunet_head_gn_bwd (the last
  // activation and the head's input gradient are never written) -- only in that fused form (32 / 64 head channels, 2..4 classes)
  n->gn_head = norm == 1 && !env_on("IUNET_NO_HEAD_ACT") && !env_on("IUNET_NO_HEAD_BN_FUSION") && iunet_head_bn_bwd_ok(base, ncls);
  n->head_bn = n->head_act && !env_on("IUNET_NO_HEAD_BN_FUSION");          // head backward + the last conv's BatchNorm backward in two passes over y (iunet_head_bn_bwd)
  n->gn_conv_stats = norm == 1 && !env_on("IUNET_NO_GN_CONV_STATS");
  n->gn_bw = n->gn_conv_stats && !env_on("IUNET_NO_GN_BW_FUSION");      // ... and the backward's sums from the data gradient's epilogue, per sample      // GroupNorm statistics from the conv epilogue (per sample) where the launch has that form
  for (int l = 0; l < levels; ++l) n->ch.push_back(base << l);
  long long off = 0, pk = 0;
  int nbn = 0;
  auto add = [&](const std::string& name, long long numel) { n->params.push_back({name, off, numel}); const long long o = off; off += numel; return o; };
  auto stage = [&](const std::string& prefix, int ci, int co, int l) {
    for (int j = 1; j <= 2; ++j) {
      TConv c;
      c.ci = j == 1 ? ci : co; c.co = co; c.l = l; c.first = (prefix == "enc0" && j == 1);
      c.name = prefix + ".conv" + std::to_string(j);
      const std::string b = prefix + ".bn" + std::to_string(j);
      c.w = add(c.name + ".weight", (long long)co * c.ci * n->taps);
      c.gamma = add(b + ".weight", co);
      c.beta = add(b + ".bias", co);
      c.bn = nbn++;
      if (c.first) {
        c.first_elems = iunet_pack_first_conv_elems(co, c.ci, n->taps);
        c.first_pk = pk; pk = align256(pk + c.first_elems * 2);
        n->ndesc += 1;
      } else {
        pack_alloc(c.fwd, co, c.ci, n->taps, 0, pk);
        pack_alloc(c.dgr, co, c.ci, n->taps, 1, pk);
        for (int k = 0; k < 4; ++k) n->ndesc += (c.fwd.buf[k] >= 0) + (c.dgr.buf[k] >= 0);
      }
      n->conv.push_back(c);
    }
  };
  for (int l = 0; l < levels; ++l) stage("enc" + std::to_string(l), l == 0 ? cin : n->ch[l - 1], n->ch[l], l);
  for (int l = levels - 2; l >= 0; --l) {
    TUp u;
    u.ci = n->ch[l + 1]; u.co = n->ch[l]; u.l = l;
    const std::string p = "dec" + std::to_string(l);
    u.w = add(p + ".up.weight", (long long)u.ci * u.co * n->npos);
    u.b = add(p + ".up.bias", u.co);
    const long long ne = (long long)u.ci * u.co * n->npos;
    u.fwd = pk; pk = align256(pk + ne * 2);
    u.dgr = pk; pk = align256(pk + ne * 2);
    n->ndesc += 2;
    n->up.push_back(u);
    stage(p, 2 * n->ch[l], n->ch[l], l);
  }
  n->head_w = add("head.weight", (long long)ncls * n->ch[0]);
  n->head_b = add("head.bias", ncls);
  n->nparams = off;
  n->table_off = pk;
  n->packed_bytes = pk + align256((long long)n->ndesc * sizeof(PackDesc));
  n->running.assign(2 * nbn, nullptr);
  *out = n;
  return IUNET_OK;
}

void iunet_train_destroy(iunet_train* n) { delete n; }

/* the flat fp32 vectors (parameters, gradient, both AdamW moments): the TRAINABLE tensors in the canonical order -- the state_dict keys
 * of interactive_unet/unet.py without the BatchNorm running statistics */
long long iunet_train_num_params(const iunet_train* n) { return n ? n->nparams : 0; }
int iunet_train_num_tensors(const iunet_train* n) { return n ? (int)n->params.size() : 0; }
int iunet_train_param(const iunet_train* n, int index, char* name, int name_cap, long long* offset, long long* numel) {
  IUNET_REQUIRE(n != nullptr, "train_param: null handle");
  IUNET_REQUIRE(index >= 0 && index < (int)n->params.size(), "train_param: index %d out of range", index);
  const TParam& p = n->params[index];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", p.name.c_str());
  if (offset) *offset = p.off;
  if (numel) *numel = p.numel;
  return IUNET_OK;
}
/* BatchNorm layers in canonical order (enc0.bn1, enc0.bn2, ..., dec{L-2}.bn1, ...): iunet_train_bind takes two pointers per layer */
int iunet_train_num_bn(const iunet_train* n) { return n ? (int)n->running.size() / 2 : 0; }
long long iunet_train_packed_bytes(const iunet_train* n) { return n ? n->packed_bytes : 0; }
long long iunet_train_workspace_bytes(const iunet_train* n, int N, int D, int H, int W) {
  if (!n || N < 1 || D < 1 || H < 1 || W < 1) return 0;
  const int f = 1 << (n->levels - 1);
  if (H % f || W % f || (n->dim == 3 && D % f) || (n->dim == 2 && D != 1)) return 0;
  return ws_layout(n, N, D, H, W).bytes;
}

/* Bind the caller's device buffers: flat / grad / m / v (iunet_train_num_params floats each), running = 2 * iunet_train_num_bn device
 * pointers (running_mean, running_var of each BatchNorm, canonical order; the array itself is host memory), packed
 * (iunet_train_packed_bytes bytes), state (8 x 4 bytes, iunet_train_state_init).  Writes the re-pack descriptor table into `packed` and
 * packs the operators from `flat` (one launch). */
int iunet_train_bind(iunet_train* n, void* flat, void* grad, void* m, void* v, void* const* running, void* packed, void* state, void* stream) {
  IUNET_REQUIRE(n && flat && grad && m && v && running && packed && state, "train_bind: null pointer");
  n->flat = (float*)flat; n->grad = (float*)grad; n->m = (float*)m; n->v = (float*)v; n->packed = (unsigned char*)packed;
  n->state = (float*)state;
  for (size_t i = 0; i < n->running.size(); ++i) {
    IUNET_REQUIRE(running[i] != nullptr, "train_bind: null running-statistics pointer %d", (int)i);
    n->running[i] = (float*)running[i];
  }
  std::vector<PackDesc> descs;
  auto desc = [&](const float* w, void* dst, long long total, int cout, int cin, int taps, int kind, int dg) {
    PackDesc d{};
    d.w = w; d.dst = dst; d.total = total; d.Cout = cout; d.Cin = cin; d.taps = taps; d.kind = kind; d.dgrad = dg; d.dtype = n->dtype;
    d.eps = 1e-5f;
    descs.push_back(d);
  };
  auto pack_descs = [&](const TPack& p, const float* w) {
    for (int lay : {3, 1, 0})
      if (p.buf[lay] >= 0) desc(w, n->packed + p.buf[lay], p.elems[lay], p.cout, p.cin, n->taps, lay == 3 ? 6 : lay == 1 ? 1 : 0, p.dg);
  };
  for (const TConv& c : n->conv) {
    if (c.first) desc(n->flat + c.w, n->packed + c.first_pk, c.first_elems, c.co, c.ci, n->taps, 2, 0);
    else { pack_descs(c.fwd, n->flat + c.w); pack_descs(c.dgr, n->flat + c.w); }
  }
  for (const TUp& u : n->up) {
    const long long ne = (long long)u.ci * u.co * n->npos;
    desc(n->flat + u.w, n->packed + u.fwd, ne, u.co, u.ci, n->npos, 3, 0);
    desc(n->flat + u.w, n->packed + u.dgr, ne, u.co, u.ci, n->npos, 4, 0);
  }
  IUNET_REQUIRE((int)descs.size() == n->ndesc, "train_bind: descriptor count %d != %d", (int)descs.size(), n->ndesc);
  IUNET_CHECK_HIP(hipMemcpyAsync(n->packed + n->table_off, descs.data(), descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice, (hipStream_t)stream));
  IUNET_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));        // (the table is a host vector about to go away; binding is not on the hot path)
  return iunet_pack_batch(n->packed + n->table_off, n->ndesc, 0, stream);
}

/* fp32 master weights -> packed operators (after the parameters changed outside iunet_train_update) */
int iunet_train_repack(iunet_train* n, void* stream) {
  IUNET_REQUIRE(n && n->packed, "train_repack: iunet_train_bind has not been called");
  return iunet_pack_batch(n->packed + n->table_off, n->ndesc, 0, stream);
}

/* unet.py:88-102 + the backward pass: X = the caller's tensor (in_dtype 0 f32, 1 f16, 2 u8 scaled by 1 / 255, 3 bf16; element strides n,
 * c, d, h, w), target / weight = [N][ncls][D*H*W] of tdtype (0 f32, 1 f16; weight may be null -- loader.py:142-154's batch contract).
 * Leaves loss_scale x dLoss/dparameter in the bound gradient vector, out4 = [Loss, Dice, IoU, MCC] (fp32 device pointer, optional),
 * the BatchNorm running statistics updated (momentum 0.1).  workspace: iunet_train_workspace_bytes. */
typedef void (*iunet_train_hook)(void* ctx, int stage);
int iunet_train_forward_backward_hooks(iunet_train* n, const void* x, int in_dtype, const long long* in_strides, const void* target,
                                       const void* weight, int tdtype, int N, int D, int H, int W, void* workspace, void* out4,
                                       iunet_train_hook hook, void* hook_ctx, void* stream);
int iunet_train_forward_backward(iunet_train* n, const void* x, int in_dtype, const long long* in_strides, const void* target,
                                 const void* weight, int tdtype, int N, int D, int H, int W, void* workspace, void* out4, void* stream) {
  return iunet_train_forward_backward_hooks(n, x, in_dtype, in_strides, target, weight, tdtype, N, D, H, W, workspace, out4, nullptr, nullptr, stream);
}

/* The same with a HOST callback between the backward's launches, for data parallelism (trainer.py:56-63 on N GPUs; one process per GPU):
 * hook(ctx, 0) is called when every launch that writes the DECODER + HEAD gradients (the tail of the flat vector, from the first
 * decoder parameter on) has been enqueued, hook(ctx, 1) when the BOTTOM encoder level's are -- the caller starts the all-reduce of that
 * bucket there (behind an event on `stream`), so that it travels over xGMI while the rest of the backward computes; the remaining head
 * of the vector is complete when the call returns.  hook may be null. */
int iunet_train_forward_backward_hooks(iunet_train* n, const void* x, int in_dtype, const long long* in_strides, const void* target,
                                       const void* weight, int tdtype, int N, int D, int H, int W, void* workspace, void* out4,
                                       iunet_train_hook hook, void* hook_ctx, void* stream) {
  IUNET_REQUIRE(n && x && in_strides && target && workspace, "train_forward_backward: null pointer");
  IUNET_REQUIRE(n->packed != nullptr, "train_forward_backward: iunet_train_bind has not been called");
  IUNET_REQUIRE(iunet_train_workspace_bytes(n, N, D, H, W) > 0, "train_forward_backward: spatial size %d x %d x %d must be divisible by %d (D == 1 in 2-D)",
                D, H, W, 1 << (n->levels - 1));
  IUNET_REQUIRE(tdtype == 0 || tdtype == 1, "train_forward_backward: target dtype must be 0 (f32) or 1 (f16)");
  const TWs L = ws_layout(n, N, D, H, W);
  unsigned char* WS = (unsigned char*)workspace;
  unsigned char* K = n->packed;
  float* P = n->flat;
  float* G = n->grad;
  const int lv = n->levels, dim = n->dim, dt = n->dtype;
  const float eps = 1e-5f, momentum = 0.1f;
  auto dims = [&](int l, int& d, int& h, int& w) { d = dim == 3 ? D >> l : 1; h = H >> l; w = W >> l; };
  auto vox = [&](int l) { int d, h, w; dims(l, d, h, w); return (long long)d * h * w; };
  auto idx = [&](bool dec, int l, int j) { return 2 * stage_index(n, dec, l) + (j - 1); };
  auto F = [&](long long off) { return (float*)(WS + off); };
  int rc = 0;
  std::vector<int> bw_ready(n->conv.size(), -1);

  // ---- forward of one stage conv: raw output y + BatchNorm batch statistics -> scale / shift; z = relu(bn(y)) unless zp is null
  auto conv_fwd = [&](int k, const void* xp, long long x_ss, void* zp, long long z_ss, int x_act, void* pool_p, long long pool_ss) -> int {
    const TConv& c = n->conv[k];
    int d, h, w;
    dims(c.l, d, h, w);
    const long long v = vox(c.l);
    void* y = WS + L.y[k];
    float* stats = n->norm == 1 ? nullptr : F(L.stats);       // GroupNorm: per sample -- from the conv's epilogue where the launch has that form (gn_rows > 0), else in its own pass
    int nparts, gn_rows = 0;
    if (c.first) {
      nparts = iunet_conv3_num_tiles(dim, N, d, h, w);
      if (n->norm == 1 && n->gn_conv_stats) { stats = F(L.stats); gn_rows = nparts / N; }      // one row per tile, a sample's tiles together: per-sample rows as they are
      rc = iunet_first_conv_fwd(dt, dim, x, in_dtype, in_strides, y, c.co * v, K + c.first_pk, nullptr, stats, N, d, h, w, c.ci, c.co, 0, stream);
    } else {
      long long woff;
      const int lay = pack_pick(c.fwd, dim, N, d, h, w, x_act >= 0, false, &woff);
      nparts = iunet_conv3_stats_parts(dim, N, d, h, w, c.co, lay);
      if (n->norm == 1 && x_act < 0 && n->gn_conv_stats) gn_rows = iunet_conv3_sample_stats_rows(dt, dim, N, d, h, w, c.ci, c.co, lay);
      if (gn_rows > 0) rc = iunet_conv3_fwd_sample_stats(dt, dim, xp, x_ss, y, c.co * v, K + woff, F(L.stats), N, d, h, w, c.ci, c.co, lay, stream);
      else if (x_act < 0) rc = iunet_conv3_fwd(dt, dim, xp, x_ss, y, c.co * v, K + woff, nullptr, stats, N, d, h, w, c.ci, c.co, 0, lay, stream);
      else rc = iunet_conv3_fwd_act(dt, dim, xp, x_ss, y, c.co * v, K + woff, nullptr, stats, F(L.scale[x_act]), F(L.shift[x_act]), N, d, h, w,
                                    c.ci, c.co, 0, lay, stream);
    }
    if (rc) return rc;
    if (n->norm == 1) {
      if (pool_p != nullptr) {
        int dn, hn, wn;
        dims(c.l + 1, dn, hn, wn);
        return iunet_gn_relu_pool_fwd_rows(dt, dim, y, c.co * v, zp, z_ss, pool_p, pool_ss, P + c.gamma, P + c.beta, n->groups, eps,
                                           gn_rows > 0 ? F(L.stats) : F(L.bnslab), gn_rows, F(L.scale[k]), F(L.shift[k]), F(L.mean[k]), F(L.invstd[k]), c.co, N,
                                           dn, hn, wn, stream);
      }
      return iunet_gn_relu_fwd_rows(dt, y, c.co * v, zp, z_ss, P + c.gamma, P + c.beta, n->groups, eps, gn_rows > 0 ? F(L.stats) : F(L.bnslab), gn_rows,
                                    F(L.scale[k]), F(L.shift[k]), F(L.mean[k]), F(L.invstd[k]), c.co, N, v, stream);
    }
    rc = iunet_bn_finalize(stats, nparts, c.co, (double)N * v, P + c.gamma, P + c.beta, n->running[2 * c.bn], n->running[2 * c.bn + 1], momentum, eps,
                           F(L.scale[k]), F(L.shift[k]), F(L.mean[k]), F(L.invstd[k]), stream);
    if (rc) return rc;
    if (zp != nullptr && pool_p != nullptr) {
      int dn, hn, wn;
      dims(c.l + 1, dn, hn, wn);
      rc = iunet_bn_relu_pool_fwd(dt, dim, y, c.co * v, zp, z_ss, pool_p, pool_ss, F(L.scale[k]), F(L.shift[k]), c.co, N, dn, hn, wn, stream);
    } else if (zp != nullptr) {
      rc = iunet_bn_relu_fwd(dt, y, c.co * v, zp, z_ss, F(L.scale[k]), F(L.shift[k]), c.co, N, v, stream);
    }
    return rc;
  };
  // input of a stage's second conv: conv1's raw output with its BatchNorm + ReLU applied by the consumers' loader waves, or the
  // materialised activation (train_engine.TrainEngine._conv2_input)
  auto conv2_fused = [&](int l) { return n->fuse_act && (dim == 3 || n->ch[l] <= 64); };

  for (int l = 0; l < lv; ++l) {
    const long long v = vox(l);
    const int c = n->ch[l], ci = l == 0 ? n->cin : n->ch[l - 1];
    const int k1 = idx(false, l, 1), k2 = idx(false, l, 2);
    const bool fused = conv2_fused(l);
    void* z1p = fused ? nullptr : (void*)(WS + L.z[k1]);
    const void* x2 = fused ? (const void*)(WS + L.y[k1]) : (const void*)(WS + L.z[k1]);
    rc = l == 0 ? conv_fwd(k1, nullptr, 0, z1p, (long long)c * v, -1, nullptr, 0)
                : conv_fwd(k1, WS + L.pin[l], (long long)ci * v, z1p, (long long)c * v, -1, nullptr, 0);
    if (rc) return rc;
    if (l < lv - 1) rc = conv_fwd(k2, x2, (long long)c * v, WS + L.cat[l], 2ll * c * v, fused ? k1 : -1, WS + L.pin[l + 1], (long long)c * vox(l + 1));
    else rc = conv_fwd(k2, x2, (long long)c * v, WS + L.z[k2], (long long)c * v, fused ? k1 : -1, nullptr, 0);
    if (rc) return rc;
  }
  for (int l = lv - 2; l >= 0; --l) {
    int di, hi, wi;
    dims(l + 1, di, hi, wi);
    const long long v = vox(l), vi = vox(l + 1);
    const int c = n->ch[l], cn = n->ch[l + 1];
    const int ksrc = l == lv - 2 ? idx(false, l + 1, 2) : idx(true, l + 1, 2);
    const TUp& u = n->up[lv - 2 - l];
    rc = iunet_convT_fwd(dt, dim, WS + L.z[ksrc], (long long)cn * vi, WS + L.cat[l] + (long long)c * v * 2, 2ll * c * v, K + u.fwd, P + u.b, N, di, hi,
                         wi, cn, c, stream);
    if (rc) return rc;
    const int k1 = idx(true, l, 1), k2 = idx(true, l, 2);
    const bool fused = conv2_fused(l);
    void* z1p = fused ? nullptr : (void*)(WS + L.z[k1]);
    const void* x2 = fused ? (const void*)(WS + L.y[k1]) : (const void*)(WS + L.z[k1]);
    rc = conv_fwd(k1, WS + L.cat[l], 2ll * c * v, z1p, (long long)c * v, -1, nullptr, 0);
    if (rc) return rc;
    // the last stage's activation is read by the head only: with head_act the head kernels apply its BatchNorm + ReLU while loading
    void* z2 = (l == 0 && (n->head_act || n->gn_head)) ? nullptr : (void*)(WS + L.z[k2]);
    rc = conv_fwd(k2, x2, (long long)c * v, z2, (long long)c * v, fused ? k1 : -1, nullptr, 0);
    if (rc) return rc;
  }

  // ---- head + softmax + loss (unet.py:88-102, metrics.py)
  const long long v0 = vox(0);
  const int c0 = n->ch[0], kl = idx(true, 0, 2);
  if (n->gn_head)
    rc = iunet_head_loss_fwd_act_ps(dt, WS + L.y[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, n->kind,
                                    F(L.lslab), F(L.out4), F(L.coef), F(L.scale[kl]), F(L.shift[kl]), 1, N, v0, stream);
  else if (n->head_act)
    rc = iunet_head_loss_fwd_act(dt, WS + L.y[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, n->kind,
                                 F(L.lslab), F(L.out4), F(L.coef), F(L.scale[kl]), F(L.shift[kl]), N, v0, stream);
  else
    rc = iunet_head_loss_fwd(dt, WS + L.z[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, n->kind,
                             F(L.lslab), F(L.out4), F(L.coef), N, v0, stream);
  if (rc) return rc;
  if (out4 != nullptr) IUNET_CHECK_HIP(hipMemcpyAsync(out4, F(L.out4), 4 * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));

  // ---- backward
  int dy_ready = -1;          // the conv whose BatchNorm backward has already run (dy holds its output gradient)
  {
    const int nparts = iunet_head_loss_bwd_num_parts(N, v0, n->ncls, c0);
    const TConv& cl = n->conv[kl];
    if (n->gn_head) {
      rc = iunet_head_gn_bwd(dt, WS + L.y[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, F(L.coef), 0.f,
                             n->state, F(L.scale[kl]), F(L.shift[kl]), F(L.mean[kl]), F(L.invstd[kl]), P + cl.gamma, n->groups, G + cl.gamma, G + cl.beta,
                             WS + L.dy, (long long)c0 * v0, F(L.hslab), F(L.bnslab), F(L.bncoef), WS + L.dz[kl], N, v0, stream);
      dy_ready = kl;
    } else if (n->head_bn && iunet_head_bn_bwd_ok(c0, n->ncls)) {
      rc = iunet_head_bn_bwd(dt, WS + L.y[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, F(L.coef), 0.f,
                             n->state, F(L.scale[kl]), F(L.shift[kl]), F(L.mean[kl]), F(L.invstd[kl]), P + cl.gamma, G + cl.gamma, G + cl.beta,
                             WS + L.dy, (long long)c0 * v0, F(L.hslab), F(L.bnslab), F(L.bncoef), WS + L.dz[kl] /* unused by this path: 32 x 2 bytes per voxel */,
                             N, v0, stream);
      dy_ready = kl;
    } else if (n->head_act)
      rc = iunet_head_loss_bwd_dev(dt, WS + L.y[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, F(L.coef),
                                   n->state, WS + L.dz[kl], (long long)c0 * v0, F(L.hslab), F(L.scale[kl]), F(L.shift[kl]), N, v0, stream);
    else
      rc = iunet_head_loss_bwd_dev(dt, WS + L.z[kl], (long long)c0 * v0, c0, P + n->head_w, P + n->head_b, n->ncls, target, weight, tdtype, F(L.coef),
                                   n->state, WS + L.dz[kl], (long long)c0 * v0, F(L.hslab), nullptr, nullptr, N, v0, stream);
    if (rc) return rc;
    rc = iunet_reduce_slab(F(L.hslab), nparts, (long long)n->ncls * (c0 + 1), F(L.htmp), 1.0f, 0, stream);
    if (rc) return rc;
    rc = iunet_head_grad_scatter(F(L.htmp), G + n->head_w, G + n->head_b, n->ncls, c0, stream);
    if (rc) return rc;
  }
  // backward of one stage conv: BatchNorm + ReLU backward, weight gradient, data gradient (train_engine.TrainEngine._stage_conv_bwd)
  auto conv_bwd = [&](int k, const void* dzp, long long dz_ss, const void* xp, long long x_ss, void* dxp, long long dx_ss, int x_act,
                      const void* dpool, long long dpool_ss, int feeds) -> int {
    const TConv& c = n->conv[k];
    int d, h, w;
    dims(c.l, d, h, w);
    const long long v = vox(c.l);
    void* dy = WS + L.dy;
    const void* y = WS + L.y[k];
    if (k == dy_ready) {
      // (iunet_head_bn_bwd has written dy, dgamma and dbeta of this conv)
    } else if (n->norm == 1 && dpool != nullptr) {
      int dn, hn, wn;
      dims(c.l + 1, dn, hn, wn);
      rc = iunet_gn_relu_pool_bwd(dt, dim, dzp, dz_ss, dpool, dpool_ss, y, c.co * v, dy, c.co * v, P + c.gamma, n->groups, F(L.scale[k]), F(L.shift[k]),
                                  F(L.mean[k]), F(L.invstd[k]), G + c.gamma, G + c.beta, F(L.bnslab), F(L.bncoef), c.co, N, dn, hn, wn, stream);
    } else if (n->norm == 1) {
      // (bw_ready: the data-gradient launch that produced dz left this layer's per-sample sums in L.stats -- no reduction pass)
      const int rows = bw_ready[k] > 0 ? bw_ready[k] : 0;
      bw_ready[k] = -1;
      rc = iunet_gn_relu_bwd_rows(dt, dzp, dz_ss, y, c.co * v, dy, c.co * v, P + c.gamma, n->groups, F(L.scale[k]), F(L.shift[k]), F(L.mean[k]), F(L.invstd[k]),
                                  G + c.gamma, G + c.beta, rows > 0 ? F(L.stats) : F(L.bnslab), rows, F(L.bncoef), c.co, N, v, stream);
    } else if (dpool != nullptr) {
      int dn, hn, wn;
      dims(c.l + 1, dn, hn, wn);
      rc = iunet_bn_relu_pool_bwd(dt, dim, dzp, dz_ss, dpool, dpool_ss, y, c.co * v, dy, c.co * v, F(L.mean[k]), F(L.invstd[k]), P + c.gamma,
                                  F(L.scale[k]), F(L.shift[k]), G + c.gamma, G + c.beta, F(L.bnslab), F(L.bncoef), c.co, N, dn, hn, wn, stream);
    } else if (bw_ready[k] >= 0) {
      const int nparts = bw_ready[k];
      bw_ready[k] = -1;
      rc = iunet_bn_relu_bwd_apply(dt, dzp, dz_ss, y, c.co * v, c.first ? nullptr : dy, c.co * v, F(L.mean[k]), F(L.invstd[k]), P + c.gamma,
                                   F(L.scale[k]), F(L.shift[k]), G + c.gamma, G + c.beta, F(L.stats), nparts, F(L.bncoef), c.co, N, v, stream);
    } else {
      rc = iunet_bn_relu_bwd(dt, dzp, dz_ss, nullptr, dz_ss, y, c.co * v, c.first ? nullptr : dy, c.co * v, F(L.mean[k]), F(L.invstd[k]), P + c.gamma,
                             F(L.scale[k]), F(L.shift[k]), G + c.gamma, G + c.beta, F(L.bnslab), F(L.bncoef), c.co, N, v, stream);
    }
    if (rc) return rc;
    float* gw = G + c.w;
    if (c.first && n->norm == 1)
      return iunet_first_conv_wgrad(dt, dim, x, in_dtype, in_strides, dy, c.co * v, F(L.wslab), gw, N, d, h, w, c.ci, c.co, stream);
    if (c.first)
      return iunet_first_conv_wgrad_bn(dt, dim, x, in_dtype, in_strides, dzp, dz_ss, y, c.co * v, F(L.mean[k]), F(L.invstd[k]), F(L.bncoef),
                                       F(L.scale[k]), F(L.shift[k]), F(L.wslab), gw, N, d, h, w, c.ci, c.co, stream);
    if (x_act < 0) rc = iunet_conv3_wgrad(dt, dim, xp, x_ss, dy, c.co * v, F(L.wslab), gw, 1.0f, N, d, h, w, c.ci, c.co, stream);
    else rc = iunet_conv3_wgrad_act(dt, dim, xp, x_ss, dy, c.co * v, F(L.wslab), gw, 1.0f, F(L.scale[x_act]), F(L.shift[x_act]), N, d, h, w, c.ci,
                                    c.co, stream);
    if (rc) return rc;
    long long woff;
    const int lay = pack_pick(c.dgr, dim, N, d, h, w, false, feeds >= 0 && (n->fuse_bw || n->gn_bw), &woff);      // (GroupNorm: the per-sample form of the fused sums)
    // (pack_pick keeps the request for the fused sums only where the launch has them: layout 2, or the compact operator in 2-D up to 64 channels)
    if (feeds >= 0 && n->fuse_bw && (lay == 2 || (lay == 3 && iunet_conv3_compact_ok(dim, N, d, h, w, c.co, c.ci, 0, 1)))) {
      rc = iunet_conv3_dgrad_bnstats_lay(dt, dim, dy, c.co * v, dxp, dx_ss, K + woff, F(L.stats), WS + L.y[feeds], c.ci * v, F(L.mean[feeds]),
                                         F(L.invstd[feeds]), F(L.scale[feeds]), F(L.shift[feeds]), N, d, h, w, c.co, c.ci, lay, stream);
      bw_ready[feeds] = iunet_conv3_stats_parts(dim, N, d, h, w, c.ci, 2);
    } else {
      // GroupNorm: the same fusion per sample where the launch has that form (the parameters are [N][C] rows, the sums per sample)
      int gn_rows = 0;
      if (n->norm == 1 && feeds >= 0 && n->gn_bw && (lay == 2 || (lay == 3 && dim == 2 && c.co <= 64)))
        gn_rows = iunet_conv3_sample_stats_rows(dt, dim, N, d, h, w, c.co, c.ci, lay);
      if (gn_rows > 0) {
        rc = iunet_conv3_dgrad_sample_bnstats(dt, dim, dy, c.co * v, dxp, dx_ss, K + woff, F(L.stats), WS + L.y[feeds], c.ci * v, F(L.mean[feeds]),
                                              F(L.invstd[feeds]), F(L.scale[feeds]), F(L.shift[feeds]), N, d, h, w, c.co, c.ci, lay, stream);
        bw_ready[feeds] = gn_rows;
      } else {
        rc = iunet_conv3_fwd(dt, dim, dy, c.co * v, dxp, dx_ss, K + woff, nullptr, nullptr, N, d, h, w, c.co, c.ci, 0, lay, stream);
      }
    }
    return rc;
  };
  // decoder, level 0 upwards
  for (int l = 0; l < lv - 1; ++l) {
    int di, hi, wi;
    dims(l + 1, di, hi, wi);
    const long long v = vox(l), vi = vox(l + 1);
    const int c = n->ch[l], cn = n->ch[l + 1];
    const int k1 = idx(true, l, 1), k2 = idx(true, l, 2);
    const bool fused = conv2_fused(l);
    const void* x2 = fused ? (const void*)(WS + L.y[k1]) : (const void*)(WS + L.z[k1]);
    rc = conv_bwd(k2, WS + L.dz[k2], (long long)c * v, x2, (long long)c * v, WS + L.dz[k1], (long long)c * v, fused ? k1 : -1, nullptr, 0, k1);
    if (rc) return rc;
    rc = conv_bwd(k1, WS + L.dz[k1], (long long)c * v, WS + L.cat[l], 2ll * c * v, WS + L.dcat[l], 2ll * c * v, -1, nullptr, 0, -1);
    if (rc) return rc;
    const int ksrc = l == lv - 2 ? idx(false, l + 1, 2) : idx(true, l + 1, 2);
    const TUp& u = n->up[lv - 2 - l];
    void* dup = WS + L.dcat[l] + (long long)c * v * 2;
    rc = iunet_convT_wgrad(dt, dim, WS + L.z[ksrc], (long long)cn * vi, dup, 2ll * c * v, F(L.wslab), F(L.bslab[l]), G + u.w, G + u.b, N, di, hi, wi, cn,
                           c, stream);
    if (rc) return rc;
    rc = iunet_convT_dgrad(dt, dim, dup, 2ll * c * v, WS + L.dz[ksrc], (long long)cn * vi, K + u.dgr, N, di, hi, wi, cn, c, stream);
    if (rc) return rc;
  }
  if (hook) hook(hook_ctx, 0);                  // decoder + head gradients enqueued
  // encoder, bottom level upwards
  for (int l = lv - 1; l >= 0; --l) {
    const long long v = vox(l);
    const int c = n->ch[l];
    const int k1 = idx(false, l, 1), k2 = idx(false, l, 2);
    const bool fused = conv2_fused(l);
    const void* x2 = fused ? (const void*)(WS + L.y[k1]) : (const void*)(WS + L.z[k1]);
    if (l == lv - 1)
      rc = conv_bwd(k2, WS + L.dz[k2], (long long)c * v, x2, (long long)c * v, WS + L.dz[k1], (long long)c * v, fused ? k1 : -1, nullptr, 0, k1);
    else      // dz = skip gradient (the skip half of dcat) + max-pool backward of dpin, formed on the fly in the BatchNorm backward
      rc = conv_bwd(k2, WS + L.dcat[l], 2ll * c * v, x2, (long long)c * v, WS + L.dz[k1], (long long)c * v, fused ? k1 : -1, WS + L.dpin[l + 1],
                    (long long)c * vox(l + 1), k1);
    if (rc) return rc;
    if (l == 0) rc = conv_bwd(k1, WS + L.dz[k1], (long long)c * v, nullptr, 0, nullptr, 0, -1, nullptr, 0, -1);
    else {
      const int cp = n->ch[l - 1];
      rc = conv_bwd(k1, WS + L.dz[k1], (long long)c * v, WS + L.pin[l], (long long)cp * v, WS + L.dpin[l], (long long)cp * v, -1, nullptr, 0, -1);
    }
    if (rc) return rc;
    if (hook && l == lv - 1 && lv > 1) hook(hook_ctx, 1);      // the bottom encoder level's gradients enqueued
  }
  return IUNET_OK;
}

/* unet.py:71-73 on the bound vectors: overflow check of the gradient (fp16), AdamW with torch's defaults unless given (b1 0.9, b2 0.999,
 * eps 1e-8, weight decay 1e-2) -- skipped, and the loss scale halved, when the gradient overflowed --, then the re-pack of the updated
 * operators.  world: ranks the bound gradient was summed over (1 for one GPU). */
int iunet_train_update(iunet_train* n, float lr, float b1, float b2, float eps, float wd, float world, void* stream) {
  IUNET_REQUIRE(n && n->packed, "train_update: iunet_train_bind has not been called");
  const int rc = iunet_adamw_step_dev(n->flat, n->grad, n->m, n->v, n->nparams, lr, b1, b2, eps, wd, n->state, n->dtype == 0 ? 1 : 0, world, stream);
  if (rc) return rc;
  return iunet_pack_batch(n->packed + n->table_off, n->ndesc, 0, stream);
}

/* one optimisation step = iunet_train_forward_backward + iunet_train_update (one GPU) */
int iunet_train_step(iunet_train* n, const void* x, int in_dtype, const long long* in_strides, const void* target, const void* weight,
                     int tdtype, int N, int D, int H, int W, void* workspace, float lr, float b1, float b2, float eps, float wd, void* out4,
                     void* stream) {
  const int rc = iunet_train_forward_backward(n, x, in_dtype, in_strides, target, weight, tdtype, N, D, H, W, workspace, out4, stream);
  if (rc) return rc;
  return iunet_train_update(n, lr, b1, b2, eps, wd, 1.0f, stream);
}

}  // extern "C"
