// Host side of libdiffsg_hip.so: builds the operator plan of UNet1D (UNetCF.py:262-356), owns the packed-weight
// arena and the fragment-layout workspaces, and enqueues the kernels of dsg_kernels.hpp / dsg_train.hpp.
// C ABI: include/diffsg.h.
#include "dsg_kernels.hpp"
#include "dsg_train.hpp"
#include "dsg_split.hpp"
#include "dsg_wide.hpp"
#include "dsg_panel.hpp"
#include "dsg_res64.hpp"
#include "dsg_tile.hpp"
#include "dsg_train_split.hpp"
#include "dsg_eval.hpp"
#include "dsg_labelgen.hpp"
#include "dsg_cogen.hpp"
#include "../../include/diffsg.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <algorithm>
#include <vector>

using namespace dsg;

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIPCK(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int pad32(int n) { return cdiv(n, 32) * 32; }
inline int groups_of(int w) { return cdiv(w, 8); }

struct Param {
    std::string name;
    long long numel;
    long long off;  // offset in the flat state-dict-order buffer (gradient bucket)
    const float* ptr = nullptr;
};

struct LinearP {           // nn.Linear
    int N = 0, K = 0;
    int w = -1, b = -1;    // param indices
};
struct NormP { int w = -1, b = -1; };

struct ResP {              // ResidualBlock (UNetCF.py:49-95)
    int in0 = 0, in1 = 0, N = 0;
    bool sclin = false;
    NormP n1, n2, n3;
    LinearP l1, l2, l3, sc, te, ce;
    int tb_off = 0;        // slice of the time table row
    size_t ce_off = 0;     // per-tile float offset of this block's precomputed condition embedding
    // packed (arena offsets, floats)
    size_t W1p, g1p, b1p, W2p, g2p, b2p, c2p, Wcp, W3p, g3p, b3p, c3p, Wscp;
    size_t W1T, W2T, W3T, WscT;  // transposed packs (data gradients)
    size_t W1h = 0, W2h = 0, W3h = 0, Wsch = 0;  // fp16-split planes (blocks >= 64 wide, sampling)
    size_t W1Th = 0, W2Th = 0, W3Th = 0, WscTh = 0;  // transposed fp16-split planes (data gradients)
    size_t Wch = 0;        // fp16-split planes of the condition embedding
    bool split = false;
    // training workspace (per-tile float offsets)
    size_t h1, h2, du1, dh1, dh2, rs1, rs2, rs3;
    size_t cs_off = 0, cs_stride = 0;   // this block's column-sum region (float offset in tr_cs) and its per-tile row length
    int gslot_out = -1, gslot_h2 = -1, gslot_h1 = -1;   // max|G| slots of dout, dh2, dh1
    long long dtb_off;     // slab scratch: dTB_b [N][T]
};

struct LinOpP {            // feature_proj / Down/Upsample / final
    LinearP l;
    NormP ln;              // final only
    bool lnact = false;
    size_t Wp, bp, gp, betap, WT;
    size_t Wh = 0;         // fp16-split planes
    size_t du, rs;         // final only (training)
};

struct TensorInfo {
    int width;
    size_t data_off, stats_off;   // forward workspace, per-tile float offsets
    size_t ga, gb;                // training workspace: gradients from the chain / skip consumer
    bool is_skip;
};

enum OpKind { OP_PROJ, OP_RES, OP_LIN, OP_FINAL };
struct Op {
    OpKind kind;
    int p;          // index into res / lin
    int in0, in1;   // tensor ids (-1: none)
    int out;        // tensor id (-1 for final)
    std::string name;
};

// launches with fewer row tiles than this leave SIMDs idle with one wave per tile (dsg_set_launch_policy)
// (narrow run: round 5 sweep, tools/policy_rows_ab.py -- at 1 536 / 2 048 tiles the small-launch form is 3-5 % of the step faster than the
// large-launch form without the LDS image: 0.489 -> 0.477 and 0.503 -> 0.479 ms per step at 24 576 / 32 768 rows; the LDS-resident form takes over above)
constexpr int kCoopMaxTilesDefault = 512, kNarrowSmallMaxTilesDefault = 2048, kCoopMaxTilesTrain = 1024;
// Persistent LDS-weight kernels (dsg_panel.hpp, dsg_res64.hpp) from this many row tiles on.  Rounds 3-4: 2 048 = 256 CUs x 8 tiles.  Round 5
// (tools/mid_batch_ab.py, same box): with the half-panel form (4-tile groups, two workgroups per CU) they beat the mid-size forms from 768
// tiles on -- 12 288 rows 0.339 -> 0.302 ms per step, 16 384: 0.356 -> 0.346, 24 576: 0.483 -> 0.454, 49 152 (3 072 tiles, where 8-tile
// groups left the second round half empty): 0.721 -> 0.674.
constexpr int kPanelMinTilesDefault = 768;

}  // namespace

constexpr int kMaxWgParts = 8;      // early parts of the weight-gradient launch (dsg_train_step)
constexpr int kWgForkMinTiles = 1024;

struct dsg_handle {
    dsg_unet_desc d;
    int td = 0;  // time_dim = 4*proj
    std::vector<Param> params;
    long long total_params = 0;
    std::vector<ResP> res;
    std::vector<LinOpP> lin;
    std::vector<TensorInfo> tensors;
    std::vector<Op> ops;
    size_t per_tile_floats = 0;
    int tb_stride = 0;
    int temb_l1w, temb_l1b, temb_l2w, temb_l2b;

    // packed-weight arena
    float* arena = nullptr;
    size_t arena_floats = 0;
    TimeBlockDesc* tdesc_dev = nullptr;
    PackDesc* pack_dev = nullptr;
    int pack_n = 0;
    long long pack_blocks = 0;
    std::vector<const float*> bound_ptrs;
    bool bound = false;

    // fp16-split path (dsg_split.hpp): wide blocks of the sampling loop
    bool use_split = true;
    // launch policy (dsg_set_launch_policy): launches of at most this many row tiles take the small-launch kernel forms
    int coop_max_tiles = kCoopMaxTilesDefault;          // wide blocks: k_resblock_c (N/32 waves per tile) and no pair kernels
    int narrow_small_max_tiles = kNarrowSmallMaxTilesDefault; // narrow run: k_fused_narrow_h<true> (first-step planes requested a stage ahead)
    // 128-wide blocks of the reverse loop: launches of at least this many tiles run the persistent panel kernels (dsg_panel.hpp:
    // one 8-wave workgroup per CU); coop_max_tiles == 0 ("every launch takes its large-launch form") lowers it to 0
    int panel_min_tiles = kPanelMinTilesDefault;
    int num_cus = 256;
    float* maxabs = nullptr;           // [params]: max|W| per tensor, refreshed at every bind
    const float** mx_ptrs_dev = nullptr; long long* mx_numel_dev = nullptr; int* mx_idx_dev = nullptr; int mx_n = 0;
    std::vector<int> mx_param;         // param index of each k_maxabs block
    PackHDesc* packh_dev = nullptr; int packh_n = 0; long long packh_blocks = 0;
    OpConstDesc* opc_desc_dev = nullptr; float* opc_dev = nullptr;   // [res | lin][4]: un-scale factors, refreshed at every bind (k_pack_h)

    // forward workspace
    int cap_rows = 0, cap_entries = 0;
    float* ws = nullptr;        // activations
    float* condfrag = nullptr;
    float* cembed = nullptr;    // [tiles_per_pass cap][ce_per_tile]: Wc silu(cond) of every block (sampling)
    float* ce_stats = nullptr;  // scratch statistics sink of the precompute launches
    size_t ce_per_tile = 0;
    size_t zero_off = 0;        // 128 zero floats in the arena
    float* tb = nullptr;        // [entries][tb_stride]
    float* st = nullptr;        // [entries][td]
    float* h1s = nullptr;       // [entries][td] Swish(lin1) of the time MLP
    float* tvals = nullptr;     // [entries]
    int* ts_ident = nullptr;    // [rows] identity index
    float* eps = nullptr;       // [2][rows][D]
    float* ywork = nullptr;     // [rows][D]
    float* freq = nullptr;      // [proj/2]
    double* red = nullptr;      // [2][kRedBlocks]
    int* step_dev = nullptr;
    double* renorm_stats = nullptr;              // dsg_set_renorm_hook: caller's 3 doubles (device), reduced across ranks by `renorm_fn`
    void (*renorm_fn)(void*) = nullptr; void* renorm_user = nullptr;
    int device = 0;              // the device that was current in dsg_create: the settings / status entry points select it themselves
    int* range_flag = nullptr;   // device word: a raw split-path operand left fp16's range since the last dsg_range_status
    int* range_pinned = nullptr; // pinned host word dsg_range_status_stream reads the flag through
    CallParams* call_dev = nullptr;
    hipStream_t cap_stream = nullptr;  // capture-only stream (the caller's may be the null stream)
    std::vector<double> op_ms;   // DSG_SAMPLE_PROFILE: summed HIP-event time per op
    bool train_prof = false;     // dsg_train_profile_enable: HIP events at the phase boundaries of dsg_train_step
    hipEvent_t tev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool tev_valid = false;
    std::vector<int> op_calls;

    // fused narrow run [fuse_lo, fuse_hi) of `ops` (inference only)
    int fuse_lo = 0, fuse_hi = 0;
    FusedOp* fused_dev = nullptr;
    std::vector<FusedOp> fused_host;
    FusedOpH* fusedh_dev = nullptr;
    // operator table of the WHOLE net for k_unet_tile (dsg_tile.hpp: one launch per pass for small batches), rebuilt with the fused
    // narrow run's table; tile_valid: the net has the shapes the kernel covers; opt_tile: dsg_set_option(DSG_OPT_TILE_STEP)
    FusedOpH* tileops_dev = nullptr;
    std::vector<FusedOpH> tileops_host;
    bool tile_valid = false, opt_tile = true;
    bool opt_panel_half = true;       // dsg_set_option(DSG_OPT_PANEL_HALF): the 128-wide panel kernels with 16 KiB panels, two workgroups per CU
    FusedOpH* fusedh_train_dev = nullptr;   // the same run for the training forward (every output stored, h1/h2 saved)
    const void* fusedh_train_key[4] = {nullptr, nullptr, nullptr, nullptr}; int fusedh_train_rows = 0;
    // inference tables (dsg_sample / dsg_unet_forward / dsg_time_op share them): what the device copy was built for
    struct FusedSig { bool valid = false, sp = false, share = false; int nrows = 0, npass = 0, uncond_tiles = 0; const void* p[8] = {}; } fused_sig;
    FusedOp* ce_dev = nullptr;          // condition-embedding Linear table (narrow blocks, one launch)
    std::vector<FusedOp> ce_host;
    CondTile* ctile_dev = nullptr; int ctile_n = 0; const float* ctile_key = nullptr; size_t ctile_cap = 0;
    std::vector<FusedOpH> fusedh_host;
    // LDS-resident form of the narrow run (k_fused_narrow_lds): per-operator LDS offsets, the copy list of every phase, the phases
    NarrowLdsOp* nlds_ops_dev = nullptr; NarrowLdsCopy* nlds_copies_dev = nullptr; uint4* nlds_image = nullptr;
    std::vector<NarrowPhaseArgs> nlds_phases;
    bool nlds_tail = false;            // the LDS form of the run also computes the 64-wide Linear at ops[fuse_hi] (kind 2)
    bool nlds_valid = false;
    // the 8-wide bottom of the net as ops [v8_lo, v8_hi) (Downsample 16 -> 8 ... Upsample 8 -> 16), or -1: computed on the vector unit
    // in float32 by the LDS form of the narrow run (dsg_narrow8.hpp); opt_v8: dsg_set_option(DSG_OPT_NARROW_VALU8)
    int v8_lo = -1, v8_hi = -1;
    bool opt_v8 = true;
    bool opt_f32_pair = true;          // exact path: block + consuming Linear in one launch (k_resblock_lin); DSG_OPT_F32_PAIR
    bool opt_time_beside = true;       // dsg_set_option(DSG_OPT_TRAIN_TIME_BESIDE): see the tail of dsg_train_step
    bool opt_wg_narrow_part = false;   // dsg_set_option(DSG_OPT_WGRAD_NARROW_PART); read when the descriptor tables are (re)built
    // the section's image in global memory (V8SecL layout: raw nn.Linear matrices and parameter vectors), gathered at every bind; the LDS
    // form of the narrow run stages it as one piece, small launches and the training forward read it from L1 / L2
    float* v8_image = nullptr; NarrowLdsCopy* v8_copies_dev = nullptr; int v8_ncopies = 0;
    int nlds_ncopies = 0;              // entries of nlds_copies_dev: dsg_bind_weights re-gathers the image from the re-packed arena
    size_t nlds_image_cap = 0;         // uint4 capacity of nlds_image (grow-only: captured graphs hold the pointer)

    // cached step graphs: per-step pair (with / without the renorm kernels), keyed by (rows, chunks); and ONE graph of a whole
    // T-step loop for short schedules, keyed by (rows, chunks, T)
    // captured reverse steps of the current workspace shape: gexec[0] = one early (renorm) step, gexec[1 + k] = 2^k later steps
    hipGraphExec_t gexec[1 + 6] = {};
    int g_rows = -1, g_chunks = -1, g_chunk_rows = -1;    // chunk_rows: the renorm kernels capture the segment size as an argument
    // chunked calls (dsg_sample_chunked): per-chunk Philox seeds and per-chunk renorm partials
    unsigned long long* seeds_dev = nullptr; int seeds_cap = 0;
    double* red_chunks = nullptr; int red_chunks_cap = 0;

    // training workspace
    size_t tr_per_tile = 0;      // floats per tile
    size_t tr_yt_frag = 0, tr_deps = 0;
    int tr_rows = 0, tr_T = 0, tr_chunks = 0;
    float* tr_ws = nullptr;
    float* tr_slabs = nullptr;   // [chunks][slab_stride]
    float* tr_gsum = nullptr;    // [slab_stride]
    unsigned* tr_gmax = nullptr; // max|G| per gradient tensor (float bits)
    unsigned* tr_gmax_t = nullptr; // [kMaxGmax][gmax_ld] per-tile words, zeroed every step
    int gmax_ld = 0;
    float* tr_cs = nullptr;      // per block: [tiles][cs_stride of the block] column sums
    int4* cs_map_dev = nullptr;  // [cs_slots] (slab destination, second destination, region base + slot, row stride)
    int cs_slots = 0;
    int n_gmax = 0;
    size_t slab_stride = 0;
    int* tr_ts = nullptr;        // [rows]
    float* tr_noise = nullptr; float* tr_mask = nullptr;   // [rows][D], [rows]: device-side draws (dsg_train_step_seeded), allocated on first use
    float* tr_yt_rm = nullptr;   // [rows][D]
    float* tr_tsave = nullptr;   // emb | h1pre | h1s | tpre | d_st | d_h1s
    long long* tw_dst_dev = nullptr; const float** tw_src_dev = nullptr; int tw_rows = 0;  // time_emb.weight rows of all blocks
    WgradDesc* wg_desc_dev = nullptr; WgradUnit* wg_unit_dev = nullptr; int wg_units = 0;
    // early parts of the weight-gradient launch: after the backward kernel of residual block number wg_forks[k] (counted from the
    // LAST block, the first one the backward reaches) the weight gradients of the blocks since the previous fork run on
    // side_stream beside the rest of the activation-gradient chain, see dsg_train_step
    std::vector<int> wg_forks = {3, 6};
    std::vector<int> wg_part_end;      // units [wg_part_end[k-1], wg_part_end[k]) = part k; the rest runs on the caller's stream
    long long* r2_dev = nullptr; int r2_n = 0; long long r2_total = 0;   // [starts | prefix]: the parameter ranges the last reduce covers
    bool r2_vec4 = false;              // ... in float4 units (k_reduce_ranges4)
    int wg_onehot_end = 0;             // ... and opens with the final part's time-table units: [wg_part_end.back(), wg_onehot_end),
    int wg_blocks_end = 0;             // then its residual blocks' other units [wg_onehot_end, wg_blocks_end), then the plain Linears'
    std::vector<int> wg_fork_ops;      // operator index after whose backward kernel part k starts
    int wg_early_lds = 40960;
    hipStream_t side_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_tail[3] = {nullptr, nullptr, nullptr};   // the tail of dsg_train_step: chain done / time path done / column sums done
    FusedBwdOpH* fbwd_dev = nullptr; int fbwd_n = 0;   // operator table of the fused narrow backward (split path), cached with the descriptors
    ColsumDesc* cs_desc_dev = nullptr; ColsumUnit* cs_unit_dev = nullptr; int cs_units = 0;
    // the descriptor tables depend only on (rows, T, precision mode) and the workspace addresses: built once, reused every step
    bool td_valid = false; int td_B = 0, td_T = 0, td_rows = 0; bool td_split = false; const void* td_key[6] = {};
};

namespace {

int add_param(dsg_handle* h, const std::string& name, long long numel) {
    Param p;
    p.name = name; p.numel = numel; p.off = h->total_params;
    h->total_params += numel;
    h->params.push_back(p);
    return (int)h->params.size() - 1;
}
LinearP add_linear(dsg_handle* h, const std::string& prefix, int K, int N) {
    LinearP l;
    l.N = N; l.K = K;
    l.w = add_param(h, prefix + ".weight", (long long)N * K);
    l.b = add_param(h, prefix + ".bias", N);
    return l;
}
NormP add_norm(dsg_handle* h, const std::string& prefix, int n) {
    NormP p;
    p.w = add_param(h, prefix + ".weight", n);
    p.b = add_param(h, prefix + ".bias", n);
    return p;
}
int add_res(dsg_handle* h, const std::string& prefix, int in0, int in1, int N) {
    ResP r;
    const int in = in0 + in1;
    r.in0 = in0; r.in1 = in1; r.N = N;
    r.n1 = add_norm(h, prefix + ".norm1", in);
    r.l1 = add_linear(h, prefix + ".lin1", in, N);
    r.n2 = add_norm(h, prefix + ".norm2", N);
    r.l2 = add_linear(h, prefix + ".lin2", N, N);
    r.n3 = add_norm(h, prefix + ".norm3", N);
    r.l3 = add_linear(h, prefix + ".lin3", N, N);
    r.sclin = in != N;
    if (r.sclin) r.sc = add_linear(h, prefix + ".shortcut", in, N);
    r.te = add_linear(h, prefix + ".time_emb", h->td, N);
    r.ce = add_linear(h, prefix + ".cond_emb", h->d.cond_dim, N);
    r.tb_off = h->tb_stride;
    h->tb_stride += pad32(N);
    h->res.push_back(r);
    return (int)h->res.size() - 1;
}
int add_tensor(dsg_handle* h, int width, bool is_skip) {
    TensorInfo t;
    t.width = width;
    t.is_skip = is_skip;
    t.data_off = h->per_tile_floats;
    h->per_tile_floats += (size_t)groups_of(width) * 256;
    t.stats_off = h->per_tile_floats;
    h->per_tile_floats += 64;
    t.ga = t.gb = 0;
    h->tensors.push_back(t);
    return (int)h->tensors.size() - 1;
}

bool width_supported(int n) { return n == 4 || n == 8 || n == 16 || n == 32 || n == 64 || n == 128; }

struct Carver {
    size_t off = 0;
    size_t take(size_t floats) { size_t o = off; off += (floats + 63) / 64 * 64; return o; }
};

void carve(dsg_handle* h) {
    Carver c;
    const int CG = groups_of(h->d.cond_dim);
    for (auto& r : h->res) {
        const int NT = cdiv(r.N, 32), NG = groups_of(r.N), KG = groups_of(r.in0) + groups_of(r.in1), OT1 = cdiv(KG, 4);
        r.W1p = c.take((size_t)NT * KG * 256);
        r.g1p = c.take((size_t)KG * 8 + 32);
        r.b1p = c.take((size_t)KG * 8 + 32);
        r.W2p = c.take((size_t)NT * NG * 256);
        r.g2p = c.take(NT * 32); r.b2p = c.take(NT * 32); r.c2p = c.take(NT * 32);
        r.Wcp = c.take((size_t)NT * CG * 256);
        r.W3p = c.take((size_t)NT * NG * 256);
        r.g3p = c.take(NT * 32); r.b3p = c.take(NT * 32); r.c3p = c.take(NT * 32);
        r.Wscp = r.sclin ? c.take((size_t)NT * KG * 256) : 0;
        r.W1T = c.take((size_t)OT1 * NG * 256);
        r.W2T = c.take((size_t)NT * NG * 256);
        r.W3T = c.take((size_t)NT * NG * 256);
        r.WscT = r.sclin ? c.take((size_t)OT1 * NG * 256) : 0;
        r.split = true;
        if (r.split) {
            const size_t KS1 = (size_t)(groups_of(r.in0) + 1) / 2 + (size_t)(groups_of(r.in1) + 1) / 2;
            r.W1h = c.take((size_t)NT * KS1 * 128 * 4);
            r.W2h = c.take((size_t)NT * ((NG + 1) / 2) * 128 * 4);
            r.W3h = c.take((size_t)NT * ((NG + 1) / 2) * 128 * 4);
            r.Wsch = r.sclin ? c.take((size_t)NT * KS1 * 128 * 4) : 0;
            const size_t KSn = (size_t)(NG + 1) / 2;
            r.W1Th = c.take((size_t)OT1 * KSn * 128 * 4);
            r.W2Th = c.take((size_t)NT * KSn * 128 * 4);
            r.W3Th = c.take((size_t)NT * KSn * 128 * 4);
            r.WscTh = r.sclin ? c.take((size_t)OT1 * KSn * 128 * 4) : 0;
        }
    }
    for (auto& l : h->lin) {
        const int NT = cdiv(l.l.N, 32), KG = groups_of(l.l.K);
        l.Wp = c.take((size_t)NT * KG * 256);
        l.bp = c.take(NT * 32);
        l.gp = l.lnact ? c.take((size_t)KG * 8 + 32) : 0;
        l.betap = l.lnact ? c.take((size_t)KG * 8 + 32) : 0;
        l.WT = c.take((size_t)cdiv(KG, 4) * groups_of(l.l.N) * 256);
        l.Wh = c.take((size_t)NT * ((KG + 1) / 2) * 128 * 4);
    }
    {   // condition-embedding planes of every block, consecutive: one grouped launch walks them (k_cond_embed_h)
        const size_t KSc = (size_t)(CG + 1) / 2;
        for (auto& r : h->res) r.Wch = c.take((size_t)cdiv(r.N, 32) * KSc * 128 * 4);
    }
    h->zero_off = c.take(128);
    h->arena_floats = c.off;
    for (auto& r : h->res) { r.ce_off = h->ce_per_tile; h->ce_per_tile += (size_t)groups_of(r.N) * 256; }
    // training workspace layout (per-tile float offsets)
    size_t o = 0;
    auto take = [&](size_t f) { size_t r = o; o += f; return r; };
    for (auto& t : h->tensors) {
        t.ga = take((size_t)groups_of(t.width) * 256);
        t.gb = t.is_skip ? take((size_t)groups_of(t.width) * 256) : 0;
    }
    for (auto& r : h->res) {
        const size_t NGf = (size_t)groups_of(r.N) * 256, KGf = (size_t)(groups_of(r.in0) + groups_of(r.in1)) * 256;
        r.h1 = take(NGf); r.h2 = take(NGf); r.du1 = take(KGf);
        r.dh1 = take(NGf); r.dh2 = take(NGf); r.rs1 = take(64); r.rs2 = take(64); r.rs3 = take(64);
    }
    for (auto& l : h->lin)
        if (l.lnact) { l.du = take((size_t)groups_of(l.l.K) * 256); l.rs = take(64); }
    h->tr_yt_frag = take((size_t)groups_of(h->d.input_dim) * 256);
    h->tr_deps = take((size_t)groups_of(h->d.input_dim) * 256);
    h->tr_per_tile = o;
}

void free_graphs(dsg_handle* h) {
    for (auto& g : h->gexec)
        if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    h->g_rows = h->g_chunks = h->g_chunk_rows = -1;
}

constexpr int kMaxGmax = 256;   // distinct gradient tensors whose max|G| is tracked (3 per block + 1 per Linear)

void free_train_workspace(dsg_handle* h) {
    void* ptrs[] = {h->tr_ws, h->tr_slabs, h->tr_gsum, h->tr_ts, h->tr_noise, h->tr_mask, h->tr_yt_rm, h->tr_tsave, h->wg_desc_dev, h->wg_unit_dev,
                    h->cs_desc_dev, h->cs_unit_dev, h->tr_gmax, h->tr_gmax_t, h->tr_cs, h->cs_map_dev};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->tr_ws = h->tr_slabs = h->tr_gsum = h->tr_yt_rm = h->tr_tsave = nullptr;
    h->tr_ts = nullptr; h->tr_noise = nullptr; h->tr_mask = nullptr; h->tr_gmax = nullptr; h->tr_gmax_t = nullptr; h->tr_cs = nullptr; h->cs_map_dev = nullptr;
    h->wg_desc_dev = nullptr; h->wg_unit_dev = nullptr; h->cs_desc_dev = nullptr; h->cs_unit_dev = nullptr;
    h->wg_units = h->cs_units = 0;
    h->tr_rows = h->tr_T = 0;
    h->td_valid = false;
}

void free_workspace(dsg_handle* h) {
    h->fused_sig.valid = false;
    free_graphs(h);
    free_train_workspace(h);  // its descriptors point into the forward workspace
    void* ptrs[] = {h->ws, h->condfrag, h->tb, h->st, h->tvals, h->ts_ident, h->eps, h->ywork, h->cembed, h->ce_stats, h->h1s};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h->cembed = h->ce_stats = h->h1s = nullptr;
    h->ws = h->condfrag = h->tb = h->st = h->tvals = h->eps = h->ywork = nullptr;
    h->ts_ident = nullptr;
    h->cap_rows = h->cap_entries = 0;
}

__global__ void k_iota(int* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
__global__ void k_linspace_t(float* p, int T) {  // t = i / T in float32, as `torch.full(..., i) / T` (MSR.py:126)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) p[i] = (float)i / (float)T;
}

int ensure_workspace(dsg_handle* h, int rows, int entries) {
    if (rows <= h->cap_rows && entries <= h->cap_entries) return 0;
    const int nrows = rows > h->cap_rows ? rows : h->cap_rows;
    // the per-entry tables are a few MB: size them for 1024 schedule entries at once (T = 1000 is the longest shipped schedule), so
    // that a longer call after a short one -- a K-step run after a W-step warm-up -- does not free and re-create the multi-GB row
    // workspace and the captured graphs in the middle of a timed region
    int nent = entries > h->cap_entries ? entries : h->cap_entries;
    if (nent < 1024) nent = 1024;
    HIPCK(hipDeviceSynchronize());
    free_workspace(h);
    const size_t tiles = (size_t)cdiv(nrows, 32) * 2;  // two passes
    const int D = h->d.input_dim, CG = groups_of(h->d.cond_dim);
    HIPCK(hipMalloc(&h->ws, tiles * h->per_tile_floats * sizeof(float)));
    HIPCK(hipMemset(h->ws, 0, tiles * h->per_tile_floats * sizeof(float)));
    HIPCK(hipMalloc(&h->condfrag, (tiles / 2) * CG * 256 * sizeof(float)));
    HIPCK(hipMalloc(&h->cembed, (tiles / 2) * h->ce_per_tile * sizeof(float)));
    HIPCK(hipMalloc(&h->ce_stats, (tiles / 2) * 64 * sizeof(float)));
    HIPCK(hipMalloc(&h->tb, (size_t)nent * h->tb_stride * sizeof(float)));
    HIPCK(hipMalloc(&h->st, (size_t)nent * h->td * sizeof(float)));
    HIPCK(hipMalloc(&h->h1s, (size_t)nent * h->td * sizeof(float)));
    HIPCK(hipMalloc(&h->tvals, (size_t)nent * sizeof(float)));
    HIPCK(hipMalloc(&h->ts_ident, (size_t)nrows * sizeof(int)));
    HIPCK(hipMalloc(&h->eps, (size_t)2 * nrows * D * sizeof(float)));
    HIPCK(hipMalloc(&h->ywork, (size_t)nrows * D * sizeof(float)));
    hipLaunchKernelGGL(k_iota, dim3(cdiv(nrows, 256)), dim3(256), 0, 0, h->ts_ident, nrows);
    HIPCK(hipDeviceSynchronize());
    h->cap_rows = nrows;
    h->cap_entries = nent;
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// kernel dispatch
// ------------------------------------------------------------------------------------------------------
template <int N>
void launch_res_n(bool sclin, const BlockArgs& a, hipStream_t s) {
    const dim3 grid(cdiv(a.ntiles, kWavesPerBlock)), block(256);
    if constexpr (N >= 64) {      // the unrolled-chain form when the input tensors have N / 8 groups each (every shipped net)
        if (a.in0.groups == N / 8 && a.in1.groups == (sclin ? N / 8 : 0)) {
            if (sclin) hipLaunchKernelGGL((k_resblock<N, true, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_resblock<N, false, true>), grid, block, 0, s, a);
            return;
        }
    }
    if (sclin) hipLaunchKernelGGL((k_resblock<N, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_resblock<N, false>), grid, block, 0, s, a);
}
void launch_res(int N, bool sclin, const BlockArgs& a, hipStream_t s) {
    switch (N) {
        case 4: launch_res_n<4>(sclin, a, s); break;
        case 8: launch_res_n<8>(sclin, a, s); break;
        case 16: launch_res_n<16>(sclin, a, s); break;
        case 32: launch_res_n<32>(sclin, a, s); break;
        case 64: launch_res_n<64>(sclin, a, s); break;
        case 128: launch_res_n<128>(sclin, a, s); break;
    }
}
template <int N>
void launch_res_bwd_n(bool sclin, const BlockBwdArgs& a, hipStream_t s) {
    const dim3 grid(cdiv(a.ntiles, kWavesPerBlock)), block(256);
    if (sclin) hipLaunchKernelGGL((k_resblock_bwd<N, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_resblock_bwd<N, false>), grid, block, 0, s, a);
}
template <int N>
void launch_res_bwd_h_n(bool sclin, const BlockBwdArgsH& a, hipStream_t s) {
    const dim3 grid(cdiv(a.b.ntiles, kWavesPerBlock)), block(256);
    if (sclin) hipLaunchKernelGGL((k_resblock_bwd_h<N, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_resblock_bwd_h<N, false>), grid, block, 0, s, a);
}
// the cooperative form needs the block's inputs exactly N wide (every 64- / 128-wide block of the shipped architectures)
bool bwd_coop_ok(int N, bool sclin, const BlockBwdArgsH& a) {
    if (N != 64 && N != 128) return false;
    if (a.b.in0.width != N) return false;
    return sclin ? (a.b.in1.width == N && a.WscTh != nullptr) : a.b.in1.groups == 0;
}
// `coop`: the cooperative form where the shapes allow it (faster at every batch size measured: 512 rows 72 -> 20 us per up-128 block,
// 32 768 rows 105 -> 65 us); the fused narrow backward's table keeps the one-wave-per-tile bodies
// Returns false for a shape no kernel is instantiated for: a 64- / 128-wide block whose inputs are not exactly N wide.  UNet1D builds
// none (DownBlock w -> w, UpBlock 2w -> w, UNetCF.py:278-311); the one-wave-per-tile form that covered them kept 72-108 bytes of scratch
// per lane at N = 128 and was never launched: removed rather than shipped unmeasured.
bool launch_res_bwd_h(int N, bool sclin, const BlockBwdArgsH& a, hipStream_t s, bool coop = false) {
    if ((N == 64 || N == 128) && !(coop && bwd_coop_ok(N, sclin, a))) return false;
    if (coop && bwd_coop_ok(N, sclin, a)) {
        const dim3 block(256);
        if (N == 128) {
            const dim3 grid(a.b.ntiles);
            if (sclin) hipLaunchKernelGGL((k_resblock_bwd_c<128, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_resblock_bwd_c<128, false>), grid, block, 0, s, a);
        } else {
            const dim3 grid(cdiv(a.b.ntiles, 2));
            if (sclin) hipLaunchKernelGGL((k_resblock_bwd_c<64, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_resblock_bwd_c<64, false>), grid, block, 0, s, a);
        }
        return true;
    }
    switch (N) {
        case 4: launch_res_bwd_h_n<4>(sclin, a, s); break;
        case 8: launch_res_bwd_h_n<8>(sclin, a, s); break;
        case 16: launch_res_bwd_h_n<16>(sclin, a, s); break;
        case 32: launch_res_bwd_h_n<32>(sclin, a, s); break;
        default: return false;
    }
    return true;
}
void launch_res_bwd(int N, bool sclin, const BlockBwdArgs& a, hipStream_t s) {
    switch (N) {
        case 4: launch_res_bwd_n<4>(sclin, a, s); break;
        case 8: launch_res_bwd_n<8>(sclin, a, s); break;
        case 16: launch_res_bwd_n<16>(sclin, a, s); break;
        case 32: launch_res_bwd_n<32>(sclin, a, s); break;
        case 64: launch_res_bwd_n<64>(sclin, a, s); break;
        case 128: launch_res_bwd_n<128>(sclin, a, s); break;
    }
}
template <int NT>
void launch_lin_nt(int inmode, int outmode, bool lnact, const LinArgs& a, hipStream_t s) {
    const dim3 grid(cdiv(a.ntiles, kWavesPerBlock)), block(256);
    if (inmode == IN_ROWMAJOR) hipLaunchKernelGGL((k_linear<NT, IN_ROWMAJOR, OUT_FRAG, false>), grid, block, 0, s, a);
    else if (outmode == OUT_ROWMAJOR && lnact) hipLaunchKernelGGL((k_linear<NT, IN_FRAG, OUT_ROWMAJOR, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_linear<NT, IN_FRAG, OUT_FRAG, false>), grid, block, 0, s, a);
}
void launch_lin(int N, int inmode, int outmode, bool lnact, const LinArgs& a, hipStream_t s) {
    switch (cdiv(N, 32)) {
        case 1: launch_lin_nt<1>(inmode, outmode, lnact, a, s); break;
        case 2: launch_lin_nt<2>(inmode, outmode, lnact, a, s); break;
        case 3: launch_lin_nt<3>(inmode, outmode, lnact, a, s); break;
        case 4: launch_lin_nt<4>(inmode, outmode, lnact, a, s); break;
    }
}
template <int OT>
void launch_lin_bwd_ot(bool lnbwd, const LinBwdArgs& a, hipStream_t s) {
    const dim3 grid(cdiv(a.ntiles, kWavesPerBlock)), block(256);
    if (lnbwd) hipLaunchKernelGGL((k_linear_bwd<OT, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_linear_bwd<OT, false>), grid, block, 0, s, a);
}
void launch_lin_bwd(int K, bool lnbwd, const LinBwdArgs& a, hipStream_t s) {
    switch (cdiv(K, 32)) {
        case 1: launch_lin_bwd_ot<1>(lnbwd, a, s); break;
        case 2: launch_lin_bwd_ot<2>(lnbwd, a, s); break;
        case 3: launch_lin_bwd_ot<3>(lnbwd, a, s); break;
        case 4: launch_lin_bwd_ot<4>(lnbwd, a, s); break;
    }
}

struct RunCtx {
    int nrows;           // rows per pass
    int npass;           // 1 or 2
    int uncond_tiles;    // leading tiles without the condition term
    const float* y;      // row-major [nrows][D]
    float* eps_out;      // row-major [npass][nrows][D]
    const int* step_ptr; // or null
    const int* ts;       // or null
    bool train;          // save h1/h2 for the backward pass
    bool cond_pre;       // condition embeddings precomputed for this call (dsg_sample)
    int* advance_step = nullptr;   // reverse loop: feature_proj decrements the step index (LinArgs::advance_step)
    bool share_proj = false;       // reverse loop, split path: feature_proj(y) computed for ONE pass, read by both (Seg::wrap)
};

size_t cap_tiles_of(const dsg_handle* h) { return (size_t)cdiv(h->cap_rows, 32) * 2; }
size_t tr_tiles_of(const dsg_handle* h) { return (size_t)cdiv(h->tr_rows, 32); }

Seg seg_of(const dsg_handle* h, int tid) {
    const TensorInfo& t = h->tensors[tid];
    const size_t cap = cap_tiles_of(h);
    Seg s;
    s.data = h->ws + t.data_off * cap;
    s.stats = h->ws + t.stats_off * cap;
    s.groups = groups_of(t.width);
    s.width = t.width;
    s.wrap = 0;
    return s;
}
float* trp(const dsg_handle* h, size_t off) { return h->tr_ws + off * tr_tiles_of(h); }

void fill_block_args(const dsg_handle* h, const Op& op, const RunCtx& c, BlockArgs& a) {
    const int tpp = cdiv(c.nrows, 32);
    const ResP& r = h->res[op.p];
    const float* A = h->arena;
    memset(&a, 0, sizeof a);
    a.in0 = seg_of(h, op.in0);
    if (op.in1 >= 0) a.in1 = seg_of(h, op.in1);
    if (c.share_proj) {
        const int pt = h->ops[0].out;
        if (op.in0 == pt) a.in0.wrap = tpp;
        if (op.in1 == pt) a.in1.wrap = tpp;
    }
    a.W1 = A + r.W1p; a.gamma1 = A + r.g1p; a.beta1 = A + r.b1p;
    a.tbias = h->tb + r.tb_off; a.step_ptr = c.step_ptr; a.ts = c.ts; a.tb_stride = h->tb_stride;
    a.W2 = A + r.W2p; a.gamma2 = A + r.g2p; a.beta2 = A + r.b2p; a.c2 = A + r.c2p;
    a.Wc = A + r.Wcp; a.condfrag = h->condfrag; a.cond_groups = groups_of(h->d.cond_dim);
    a.W3 = A + r.W3p; a.gamma3 = A + r.g3p; a.beta3 = A + r.b3p; a.c3 = A + r.c3p;
    a.Wsc = r.sclin ? A + r.Wscp : nullptr;
    const Seg o = seg_of(h, op.out);
    a.out = const_cast<float*>(o.data); a.out_stats = const_cast<float*>(o.stats);
    if (c.train) { a.save_h1 = trp(h, r.h1); a.save_h2 = trp(h, r.h2); }
    if (c.cond_pre) a.cond_pre = h->cembed + r.ce_off * (cap_tiles_of(h) / 2);
    a.ntiles = tpp * c.npass; a.tiles_per_pass = tpp; a.uncond_tiles = c.uncond_tiles; a.nrows = c.nrows;
    a.range_flag = h->range_flag;
    {
        const float n0 = (float)a.in0.width, n1 = (float)a.in1.width, nt = n0 + n1;
        a.chan_w = n0 * n1 / nt; a.chan_f = n1 / nt; a.inv_nin = 1.0f / nt;
    }
}

void fill_lin_args(const dsg_handle* h, const Op& op, const RunCtx& c, LinArgs& a) {
    const int tpp = cdiv(c.nrows, 32);
    const LinOpP& l = h->lin[op.p];
    const float* A = h->arena;
    memset(&a, 0, sizeof a);
    a.W = A + l.Wp; a.bias = A + l.bp;
    a.in_width = l.l.K; a.in_groups = groups_of(l.l.K);
    a.out_width = l.l.N;
    a.ntiles = tpp * c.npass; a.tiles_per_pass = tpp; a.nrows = c.nrows;
    a.range_flag = h->range_flag;
    a.inv_in_w = 1.0f / (float)l.l.K; a.inv_out_w = 1.0f / (float)l.l.N;
    if (op.kind == OP_PROJ) {
        a.in_rm = c.y; a.advance_step = c.advance_step;
        if (c.share_proj) a.ntiles = tpp;            // one pass: the second pass's consumers wrap onto these tiles
    } else {
        a.in = seg_of(h, op.in0);
        if (c.share_proj && op.in0 == h->ops[0].out) a.in.wrap = tpp;
    }
    if (op.kind == OP_FINAL) {
        a.gamma = A + l.gp; a.beta = A + l.betap; a.out_rm = c.eps_out;
    } else {
        const Seg o = seg_of(h, op.out);
        a.out = const_cast<float*>(o.data); a.out_stats = const_cast<float*>(o.stats);
    }
}

void fill_block_args_h(const dsg_handle* h, const ResP& r, const BlockArgs& b, BlockArgsH& a) {
    a.b = b;
    const float* A = h->arena;
    a.W1h = reinterpret_cast<const uint4*>(A + r.W1h); a.W2h = reinterpret_cast<const uint4*>(A + r.W2h);
    a.W3h = reinterpret_cast<const uint4*>(A + r.W3h); a.Wsch = r.sclin ? reinterpret_cast<const uint4*>(A + r.Wsch) : nullptr;
    a.m1 = h->maxabs + r.l1.w; a.m2 = h->maxabs + r.l2.w; a.m3 = h->maxabs + r.l3.w; a.msc = r.sclin ? h->maxabs + r.sc.w : nullptr;
    a.kc = h->opc_dev + 4 * (size_t)(&r - h->res.data());
}
void fill_lin_args_h(const dsg_handle* h, const LinOpP& l, const LinArgs& b, LinArgsH& a) {
    a.l = b;
    a.Wh = reinterpret_cast<const uint4*>(h->arena + l.Wh);
    a.m = h->maxabs + l.l.w;
    a.kc = h->opc_dev + 4 * (h->res.size() + (size_t)(&l - h->lin.data()));
}

template <int N>
void launch_res_h_n(bool sclin, const BlockArgsH& a, hipStream_t s) {
    const dim3 grid(cdiv(a.b.ntiles, kWavesPerBlock)), block(256);
    if (sclin) hipLaunchKernelGGL((k_resblock_h<N, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_resblock_h<N, false>), grid, block, 0, s, a);
}
// launches with fewer row tiles than this leave SIMDs idle with one wave per tile: the wide blocks then run cooperatively
// (N/32 waves per tile, k_resblock_c)

// k_wide128_h: 128-wide block whose inputs are one or two 128-wide tensors, condition embedding precomputed
bool wide128_ok(const ResP& r, const BlockArgs& b) {
    return r.N == 128 && b.in0.groups == 16 && (b.in1.groups == 0 || b.in1.groups == 16) && (b.in1.groups != 0) == r.sclin && b.cond_pre;
}

// persistent panel kernels (dsg_panel.hpp): sampling launches only (the time bias is one row: no per-row `ts`)
bool panel128_ok(const dsg_handle* h, const ResP& r, const BlockArgs& b) {
    return wide128_ok(r, b) && !b.ts && !b.save_h1 && b.tiles_per_pass * 2 >= b.ntiles && b.ntiles >= h->panel_min_tiles &&
           b.ntiles > h->coop_max_tiles;
}
template <bool SC, int EPI, int NTO>
void launch_panel128(const dsg_handle* h, const BlockLinArgsH& w, hipStream_t s) {
    if (h->opt_panel_half) {
        // half-size panels, 4 waves per workgroup, two workgroups per CU (dsg_panel.hpp, STEPS = 2)
        const int ngroups = cdiv(w.b.b.ntiles, 4);
        const dim3 grid(ngroups < 2 * h->num_cus ? ngroups : 2 * h->num_cus), block(256);
        hipLaunchKernelGGL((k_panel128_h<SC, EPI, NTO, 2>), grid, block, 0, s, w, ngroups);
        return;
    }
    const int ngroups = cdiv(w.b.b.ntiles, kPW);
    const dim3 grid(ngroups < h->num_cus ? ngroups : h->num_cus), block(kPW * 64);
    hipLaunchKernelGGL((k_panel128_h<SC, EPI, NTO>), grid, block, 0, s, w, ngroups);
}

// 64-wide blocks of large sampling launches: the block's planes resident in LDS (dsg_res64.hpp)
bool res64_lds_ok(const dsg_handle* h, const ResP& r, const BlockArgs& b) {
    return r.N == 64 && b.in0.groups == 8 && (b.in1.groups == 0 || b.in1.groups == 8) && (b.in1.groups != 0) == r.sclin && b.cond_pre && !b.ts &&
           !b.save_h1 && b.tiles_per_pass * 2 >= b.ntiles && b.ntiles >= h->panel_min_tiles && b.ntiles > h->coop_max_tiles;
}

void launch_res_h(const dsg_handle* h, const ResP& r, const BlockArgs& b, hipStream_t s) {
    BlockArgsH a;
    fill_block_args_h(h, r, b, a);
    const int ks1 = (groups_of(r.in0) + 1) / 2 + (groups_of(r.in1) + 1) / 2;
    // the cooperative body is written for inputs exactly N wide (compile-time step counts: resblock_coop_body)
    const bool coop_fits = ks1 * 128 <= kCoopLdsU4 / (4 / (r.N / 32 > 0 ? r.N / 32 : 1)) / 2 && r.in0 == r.N && r.in1 == (r.sclin ? r.N : 0);
    // the training forward (it stores h1 / h2) takes the cooperative form up to 1 024 tiles -- 32 768 rows: 1.939 -> 1.923 ms per step
    // (profiles/r04_train_tail_ab.txt); a sampling launch of that size has two passes' worth of tiles per row and stays as measured
    const int coop_max = (a.b.save_h1 && h->coop_max_tiles > 0 && h->coop_max_tiles < kCoopMaxTilesTrain) ? kCoopMaxTilesTrain : h->coop_max_tiles;
    if ((r.N == 64 || r.N == 128) && coop_fits && a.b.ntiles <= coop_max) {
        const int tpw = 4 / (r.N / 32);
        const dim3 grid(cdiv(a.b.ntiles, tpw)), block(256);
        if (r.N == 128) {
            if (r.sclin) hipLaunchKernelGGL((k_resblock_c<128, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_resblock_c<128, false>), grid, block, 0, s, a);
        } else {
            if (r.sclin) hipLaunchKernelGGL((k_resblock_c<64, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((k_resblock_c<64, false>), grid, block, 0, s, a);
        }
        return;
    }
    if (res64_lds_ok(h, r, a.b)) {
        const int ngroups = cdiv(a.b.ntiles, kR64Waves);
        const dim3 grid(ngroups < h->num_cus ? ngroups : h->num_cus), block(kR64Waves * 64);
        BlockLinArgsH w;
        memset(&w, 0, sizeof w);
        w.b = a; w.store_block_out = 1;
        if (r.sclin) hipLaunchKernelGGL((k_res64_lds<true, 0>), grid, block, 0, s, w, ngroups);
        else hipLaunchKernelGGL((k_res64_lds<false, 0>), grid, block, 0, s, w, ngroups);
        return;
    }
    if (panel128_ok(h, r, a.b)) {
        BlockLinArgsH w;
        memset(&w, 0, sizeof w);
        w.b = a; w.store_block_out = 1;
        if (r.sclin) launch_panel128<true, 0, 1>(h, w, s); else launch_panel128<false, 0, 1>(h, w, s);
        return;
    }
    if (wide128_ok(r, a.b)) {
        // large launch, 128 wide: 4 tiles per workgroup, weight planes shared through an LDS ring (dsg_wide.hpp)
        BlockLinArgsH w;
        memset(&w, 0, sizeof w);
        w.b = a; w.store_block_out = 1;
        const dim3 grid(cdiv(a.b.ntiles, 4)), block(256);
        if (r.sclin) hipLaunchKernelGGL((k_wide128_h<true, 0, 1>), grid, block, 0, s, w);
        else hipLaunchKernelGGL((k_wide128_h<false, 0, 1>), grid, block, 0, s, w);
        return;
    }
    switch (r.N) {
        case 4: launch_res_h_n<4>(r.sclin, a, s); break;
        case 8: launch_res_h_n<8>(r.sclin, a, s); break;
        case 16: launch_res_h_n<16>(r.sclin, a, s); break;
        case 32: launch_res_h_n<32>(r.sclin, a, s); break;
        case 64: launch_res_h_n<64>(r.sclin, a, s); break;
        case 128: launch_res_h_n<128>(r.sclin, a, s); break;
    }
}
template <int NT>
void launch_lin_h_nt(int inmode, int outmode, bool lnact, const LinArgsH& a, hipStream_t s) {
    const dim3 grid(cdiv(a.l.ntiles, kWavesPerBlock)), block(256);
    if (inmode == IN_ROWMAJOR) hipLaunchKernelGGL((k_linear_h<NT, IN_ROWMAJOR, OUT_FRAG, false>), grid, block, 0, s, a);
    else if (outmode == OUT_ROWMAJOR && lnact) hipLaunchKernelGGL((k_linear_h<NT, IN_FRAG, OUT_ROWMAJOR, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_linear_h<NT, IN_FRAG, OUT_FRAG, false>), grid, block, 0, s, a);
}
void launch_lin_h(int N, int inmode, int outmode, bool lnact, const LinArgsH& a, hipStream_t s) {
    switch (cdiv(N, 32)) {
        case 1: launch_lin_h_nt<1>(inmode, outmode, lnact, a, s); break;
        case 2: launch_lin_h_nt<2>(inmode, outmode, lnact, a, s); break;
        case 3: launch_lin_h_nt<3>(inmode, outmode, lnact, a, s); break;
        case 4: launch_lin_h_nt<4>(inmode, outmode, lnact, a, s); break;
    }
}

// block (64/128 wide) + the Linear that consumes it, in one launch; returns false if the shape pair is not instantiated
bool launch_res_lin_h(const dsg_handle* h, const ResP& r, const BlockArgs& b, const LinOpP& l, const LinArgs& la, bool final_op,
                      bool store_block_out, hipStream_t s) {
    BlockLinArgsH a;
    fill_block_args_h(h, r, b, a.b);
    fill_lin_args_h(h, l, la, a.l);
    a.store_block_out = store_block_out ? 1 : 0;
    a.dbg = 0;
    const dim3 grid(cdiv(b.ntiles, kWavesPerBlock)), block(256);
    const int NTO = cdiv(l.l.N, 32);
    if (res64_lds_ok(h, r, b) && l.l.K == 64 && !final_op) {
        const int ngroups = cdiv(b.ntiles, kR64Waves);
        const dim3 g64(ngroups < h->num_cus ? ngroups : h->num_cus), b64(kR64Waves * 64);
#define DSG_TRY64(SC_, NTO_)                                                                                 \
    if (r.sclin == SC_ && NTO == NTO_) {                                                                     \
        hipLaunchKernelGGL((k_res64_lds<SC_, NTO_>), g64, b64, 0, s, a, ngroups);                            \
        return true;                                                                                         \
    }
        DSG_TRY64(true, 4) DSG_TRY64(true, 2) DSG_TRY64(false, 2) DSG_TRY64(false, 1)
#undef DSG_TRY64
    }
    if (panel128_ok(h, r, b) && l.l.K == 128) {
#define DSG_TRYP(SC_, EPI_, NTO_)                                                                            \
    if (r.sclin == SC_ && (final_op ? 2 : 1) == EPI_ && NTO == NTO_) {                                       \
        launch_panel128<SC_, EPI_, NTO_>(h, a, s);                                                           \
        return true;                                                                                         \
    }
        DSG_TRYP(false, 1, 2) DSG_TRYP(false, 1, 1) DSG_TRYP(true, 2, 1) DSG_TRYP(true, 2, 3)
#undef DSG_TRYP
    }
    if (wide128_ok(r, b) && l.l.K == 128) {
#define DSG_TRYW(SC_, EPI_, NTO_)                                                                            \
    if (r.sclin == SC_ && (final_op ? 2 : 1) == EPI_ && NTO == NTO_) {                                       \
        hipLaunchKernelGGL((k_wide128_h<SC_, EPI_, NTO_>), grid, block, 0, s, a);                            \
        return true;                                                                                         \
    }
        DSG_TRYW(false, 1, 2) DSG_TRYW(false, 1, 1) DSG_TRYW(true, 2, 1) DSG_TRYW(true, 2, 3)
#undef DSG_TRYW
    }
#define DSG_TRY(N_, SC_, NTO_, FIN_)                                                                         \
    if (r.N == N_ && r.sclin == SC_ && NTO == NTO_ && final_op == FIN_) {                                    \
        hipLaunchKernelGGL((k_resblock_lin_h<N_, SC_, NTO_, FIN_>), grid, block, 0, s, a);                   \
        return true;                                                                                         \
    }
    DSG_TRY(128, false, 2, false) DSG_TRY(128, false, 1, false) DSG_TRY(64, true, 4, false) DSG_TRY(64, false, 2, false)
    DSG_TRY(64, true, 2, false) DSG_TRY(64, false, 1, false)
    DSG_TRY(128, true, 1, true) DSG_TRY(128, true, 3, true) DSG_TRY(64, true, 1, true) DSG_TRY(64, true, 3, true)
#undef DSG_TRY
    return false;
}

bool split_ctx(const dsg_handle* h, const RunCtx& c) { return h->use_split && c.cond_pre; }

void launch_op(const dsg_handle* h, const Op& op, const RunCtx& c, hipStream_t s) {
    if (op.kind == OP_RES) {
        BlockArgs a;
        fill_block_args(h, op, c, a);
        const ResP& r = h->res[op.p];
        if (split_ctx(h, c)) launch_res_h(h, r, a, s);
        else launch_res(r.N, r.sclin, a, s);
    } else {
        LinArgs a;
        fill_lin_args(h, op, c, a);
        const int inmode = op.kind == OP_PROJ ? IN_ROWMAJOR : IN_FRAG;
        const int outmode = op.kind == OP_FINAL ? OUT_ROWMAJOR : OUT_FRAG;
        if (split_ctx(h, c)) {
            LinArgsH ah;
            fill_lin_args_h(h, h->lin[op.p], a, ah);
            launch_lin_h(h->lin[op.p].l.N, inmode, outmode, op.kind == OP_FINAL, ah, s);
        } else {
            launch_lin(h->lin[op.p].l.N, inmode, outmode, op.kind == OP_FINAL, a, s);
        }
    }
}

bool fusable(const dsg_handle* h, const Op& op) {
    if (op.kind == OP_RES) return h->res[op.p].N <= 32;
    if (op.kind == OP_LIN) return h->lin[op.p].l.N <= 32;
    return false;
}

// Upload the operator descriptors of the fused narrow run for this context (stream ordered; replayed graphs read them).
int prepare_fused(dsg_handle* h, const RunCtx& c, hipStream_t s) {
    const int n = h->fuse_hi - h->fuse_lo;
    if (n < 2) return 0;
    const bool sp = split_ctx(h, c);
    if (c.train) {
        // training forward: split path only; the table depends on the workspaces and the batch, not on the step -> cached
        if (!sp) return 0;
        const void* key[4] = {h->ws, h->tr_ws, h->cembed, h->tb};
        if (h->fusedh_train_dev && h->fusedh_train_rows == c.nrows && !memcmp(key, h->fusedh_train_key, sizeof key)) return 0;
        if (!h->fusedh_train_dev) HIPCK(hipMalloc(&h->fusedh_train_dev, (h->ops.size() + 1) * sizeof(FusedOpH)));
        memcpy(h->fusedh_train_key, key, sizeof key);
        h->fusedh_train_rows = c.nrows;
    }
    if (!c.train) {
        // same context as the table already on the device (consecutive dsg_sample calls of one batch size): nothing to do - and
        // no stream synchronise between calls, so the host can enqueue the next call while this one runs
        dsg_handle::FusedSig sig;
        sig.valid = true; sig.sp = sp; sig.share = c.share_proj; sig.nrows = c.nrows; sig.npass = c.npass; sig.uncond_tiles = c.uncond_tiles;
        const void* ptrs[8] = {c.y, c.eps_out, c.step_ptr, c.ts, h->ws, h->cembed, h->tb,
                               reinterpret_cast<const void*>((size_t)h->cap_rows * 2 + (c.cond_pre ? 1 : 0))};
        memcpy(sig.p, ptrs, sizeof ptrs);
        const dsg_handle::FusedSig& o = h->fused_sig;
        if (o.valid && o.sp == sig.sp && o.share == sig.share && o.nrows == sig.nrows && o.npass == sig.npass && o.uncond_tiles == sig.uncond_tiles &&
            !memcmp(o.p, sig.p, sizeof sig.p))
            return 0;
        h->fused_sig = sig;
    }
    HIPCK(hipStreamSynchronize(s));  // the host tables may still be the source of an earlier async copy
    if (sp) h->fusedh_host.resize(n); else h->fused_host.resize(n);
    for (int i = 0; i < n; ++i) {
        const Op& op = h->ops[h->fuse_lo + i];
        BlockArgs b; LinArgs l;
        memset(&b, 0, sizeof b); memset(&l, 0, sizeof l);
        const int kind = op.kind == OP_RES ? 0 : 1;
        const int N = kind == 0 ? h->res[op.p].N : h->lin[op.p].l.N;
        const int sclin = kind == 0 ? (int)h->res[op.p].sclin : 0;
        if (kind == 0) fill_block_args(h, op, c, b); else fill_lin_args(h, op, c, l);
        if (sp) {
            FusedOpH& f = h->fusedh_host[i];
            memset(&f, 0, sizeof f);
            f.kind = kind; f.N = N; f.sclin = sclin;
            f.store_out = (c.train || h->tensors[op.out].is_skip || i == n - 1) ? 1 : 0;   // training: the backward reads every tensor
            if (kind == 0) fill_block_args_h(h, h->res[op.p], b, f.b); else fill_lin_args_h(h, h->lin[op.p], l, f.l);
        } else {
            FusedOp& f = h->fused_host[i];
            memset(&f, 0, sizeof f);
            f.kind = kind; f.N = N; f.sclin = sclin; f.b = b; f.l = l;
            // a link of the chain (bit 1): input handed over in registers; stored (bit 0) if something else reads it from memory
            f.pad = 2 | ((h->tensors[op.out].is_skip || i == n - 1) ? 1 : 0);
        }
    }
    // the Linear behind the run (Upsample 32 -> 64 of the shipped nets): the LDS form of the run computes it from the registers of
    // the run's last tensor (one launch and one round trip of that tensor less); entry n of the table, read by that form only
    bool tail = false;
    if (sp && !c.train && !c.ts && h->fuse_hi < (int)h->ops.size()) {
        const Op& t = h->ops[h->fuse_hi];
        const Op& last = h->ops[h->fuse_hi - 1];
        if (t.kind == OP_LIN && t.in0 == last.out && !h->tensors[last.out].is_skip && !h->lin[t.p].lnact && h->lin[t.p].l.N == 64 && h->lin[t.p].l.K <= 32) {
            LinArgs l;
            memset(&l, 0, sizeof l);
            fill_lin_args(h, t, c, l);
            h->fusedh_host.resize(n + 1);
            FusedOpH& f = h->fusedh_host[n];
            memset(&f, 0, sizeof f);
            f.kind = 2; f.N = 64; f.store_out = 1;
            fill_lin_args_h(h, h->lin[t.p], l, f.l);
            tail = true;
        }
    }
    if (sp) HIPCK(hipMemcpyAsync(c.train ? h->fusedh_train_dev : h->fusedh_dev, h->fusedh_host.data(), h->fusedh_host.size() * sizeof(FusedOpH), hipMemcpyHostToDevice, s));
    else HIPCK(hipMemcpyAsync(h->fused_dev, h->fused_host.data(), n * sizeof(FusedOp), hipMemcpyHostToDevice, s));
    if (sp && !c.train) {
        // the whole net's table for k_unet_tile: the run's entries as above, every other operator stored (its consumer reads memory)
        const int nops = (int)h->ops.size();
        h->tileops_host.assign(nops, FusedOpH{});
        bool ok = nops >= 2 && h->ops[0].kind == OP_PROJ && h->fuse_lo >= 1;
        int wide = 0;
        for (int i = 0; i < nops && ok; ++i) {
            const Op& op = h->ops[i];
            FusedOpH& f = h->tileops_host[i];
            memset(&f, 0, sizeof f);
            if (i >= h->fuse_lo && i < h->fuse_hi) { f = h->fusedh_host[i - h->fuse_lo]; continue; }
            f.store_out = 1;
            if (op.kind == OP_RES) {
                const ResP& r = h->res[op.p];
                BlockArgs b;
                fill_block_args(h, op, c, b);
                f.kind = 0; f.N = r.N; f.sclin = r.sclin ? 1 : 0;
                fill_block_args_h(h, r, b, f.b);
                // the cooperative body's shapes: inputs exactly N wide, images within the LDS slot (launch_res_h, coop_fits)
                const int ks1 = (groups_of(r.in0) + 1) / 2 + (groups_of(r.in1) + 1) / 2;
                if (!((r.N == 64 || r.N == 128) && r.in0 == r.N && r.in1 == (r.sclin ? r.N : 0) && ks1 <= (r.sclin ? r.N / 8 : r.N / 16) && b.cond_pre)) ok = false;
                ++wide;
            } else {
                const LinOpP& l = h->lin[op.p];
                LinArgs la;
                fill_lin_args(h, op, c, la);
                f.kind = op.kind == OP_FINAL ? 3 : 1; f.N = l.l.N;
                fill_lin_args_h(h, l, la, f.l);
                if (op.kind == OP_PROJ) f.kind = 1;            // never executed by the kernel (its range starts behind feature_proj)
                else if (l.l.N > 128 || (op.kind == OP_FINAL) != l.lnact) ok = false;
            }
        }
        h->tile_valid = ok && wide > 0;
        if (h->tile_valid)
            HIPCK(hipMemcpyAsync(h->tileops_dev, h->tileops_host.data(), (size_t)nops * sizeof(FusedOpH), hipMemcpyHostToDevice, s));
    }
    // the training forward has its own table and leaves the inference plan (and fused_sig) alone: resetting the LDS plan here
    // would drop the eager sampling path to the non-LDS kernels for good while cached graphs keep replaying the LDS form
    if (!c.train) { h->nlds_valid = false; h->nlds_tail = false; }
    if (sp && !c.train && !c.ts) {
        // LDS image of the run, cut into phases that fit kNarrowLdsU4: per operator its packed planes and per-feature vectors (static:
        // gathered once into h->nlds_image, a phase's part contiguous) and, behind them, the phase's slice of the time-table row
        std::vector<NarrowLdsOp> lops(n + 1);
        std::vector<NarrowLdsCopy> copies;                 // dst_u4: offset in the GLOBAL image buffer
        h->nlds_phases.clear();
        unsigned used = 0, image_base = 0;                 // used: static part of the current phase
        int phase_lo = 0;
        int tb_first = -1, tb_last_end = 0;                // time-table slice of the current phase (floats in the row)
        bool fits = true;
        std::vector<int> phase_of(n, 0);
        // the float32 section (ops [s_lo, s_hi) of the run): one unit of the plan, placed at its first operator
        const int s_lo = (h->opt_v8 && h->v8_lo >= 0) ? h->v8_lo - h->fuse_lo : -1, s_hi = s_lo >= 0 ? h->v8_hi - h->fuse_lo : -1;
        const int s_nb = h->d.n_blocks;
        const unsigned s_u4 = (unsigned)(s_nb == 3 ? V8SecL<3>::SIZE : V8SecL<2>::SIZE) / 4, s_tb_u4 = (unsigned)(2 * s_nb + 3) * 8;
        auto close_phase = [&](int hi) {
            NarrowPhaseArgs ph;
            ph.image = nullptr; ph.n_u4 = used; ph.tb = h->tb + (tb_first < 0 ? 0 : tb_first); ph.tb_u4 = tb_first < 0 ? 0 : (unsigned)(tb_last_end - tb_first) / 4;
            ph.op_lo = phase_lo; ph.op_hi = hi;
            ph.v8_at = -1; ph.v8_nops = 0; ph.v8_store = 0; ph.v8_sec = 0; ph.v8_tb = 0;
            if (s_lo >= phase_lo && s_lo < hi) {       // the float32 section lies in this phase (its image offset was noted when it was taken)
                ph.v8_at = s_lo; ph.v8_nops = s_hi - s_lo; ph.v8_sec = 4 * lops[s_lo].w1; ph.v8_store = lops[s_lo].store_out;
                ph.v8_tb = 4 * used + (unsigned)(h->res[h->ops[h->fuse_lo + s_lo + 1].p].tb_off - tb_first);
            }
            // image pointer filled below (needs the buffer); remember the base through n_u4 bookkeeping
            h->nlds_phases.push_back(ph);
            for (int k = phase_lo; k < hi; ++k) {           // the time-bias rows sit behind the static part
                const Op& o = h->ops[h->fuse_lo + k];
                if (o.kind == OP_RES) lops[k].tb = 4 * used + (unsigned)(h->res[o.p].tb_off - tb_first);
            }
            image_base += used;
        };
        std::vector<unsigned> phase_base;
        phase_base.push_back(0);
        for (int i = 0; i < n; ++i) {
            const Op& op = h->ops[h->fuse_lo + i];
            const float* A = h->arena;
            unsigned need = 0, need_tb = 0;
            if (s_lo >= 0 && i > s_lo && i < s_hi) {         // inside the section: planned with its first operator
                memset(&lops[i], 0, sizeof lops[i]);
                continue;
            }
            if (i == s_lo) {
                need = s_u4; need_tb = s_tb_u4;
            } else if (op.kind == OP_RES) {
                const ResP& r = h->res[op.p];
                const unsigned KG = groups_of(r.in0) + groups_of(r.in1), KS1 = (groups_of(r.in0) + 1) / 2 + (groups_of(r.in1) + 1) / 2, KS2 = (groups_of(r.N) + 1) / 2;
                need = KS1 * 128 * (r.sclin ? 2 : 1) + 2 * KS2 * 128 + 2 * ((KG * 8 + 32 + 3) / 4) + 6 * 8;
                need_tb = 8;
            } else {
                const LinOpP& l = h->lin[op.p];
                need = ((groups_of(l.l.K) + 1) / 2) * 128 + 8;
            }
            if (need + need_tb > (unsigned)kNarrowLdsU4) { fits = false; break; }
            const unsigned tb_now = tb_first < 0 ? 0 : (unsigned)(tb_last_end - tb_first) / 4;
            if (used + need + tb_now + need_tb > (unsigned)kNarrowLdsU4) {
                close_phase(i);
                // (round 6: one launch walks every phase, the running tensor crosses the boundary in registers)
                if ((int)h->nlds_phases.size() >= kNarrowMaxPhases) { fits = false; break; }
                phase_base.push_back(image_base);
                phase_lo = i; used = 0; tb_first = -1; tb_last_end = 0;
            }
            NarrowLdsOp& lo = lops[i];
            const int keep_store = lo.store_out;
            memset(&lo, 0, sizeof lo);
            lo.store_out = h->fusedh_host[(s_lo >= 0 && i == s_lo) ? s_hi - 1 : i].store_out | keep_store;
            auto take = [&](const void* src, unsigned n_u4) -> unsigned {
                const unsigned off = used;
                copies.push_back(NarrowLdsCopy{src, image_base + off, n_u4});
                used += n_u4;
                return off;
            };
            if (i == s_lo) {
                // the section's image as one piece (gathered at bind time into h->v8_image, dsg_bind_weights)
                lo.w1 = take(h->v8_image, s_u4);
                for (int k = s_lo + 1; k + 1 < s_hi; ++k) {
                    const ResP& r = h->res[h->ops[h->fuse_lo + k].p];
                    if (tb_first < 0) tb_first = r.tb_off;
                    else if (r.tb_off != tb_last_end) { fits = false; break; }
                    tb_last_end = r.tb_off + pad32(r.N);
                }
                if (!fits) break;
            } else if (op.kind == OP_RES) {
                const ResP& r = h->res[op.p];
                const unsigned KG = groups_of(r.in0) + groups_of(r.in1), KS1 = (groups_of(r.in0) + 1) / 2 + (groups_of(r.in1) + 1) / 2, KS2 = (groups_of(r.N) + 1) / 2;
                const unsigned nv1 = (KG * 8 + 32 + 3) / 4;
                lo.w1 = take(A + r.W1h, KS1 * 128);
                lo.w2 = take(A + r.W2h, KS2 * 128);
                lo.w3 = take(A + r.W3h, KS2 * 128);
                lo.wsc = r.sclin ? take(A + r.Wsch, KS1 * 128) : lo.w1;
                lo.g1 = 4 * take(A + r.g1p, nv1); lo.b1 = 4 * take(A + r.b1p, nv1);
                lo.g2 = 4 * take(A + r.g2p, 8); lo.b2 = 4 * take(A + r.b2p, 8);
                lo.g3 = 4 * take(A + r.g3p, 8); lo.b3 = 4 * take(A + r.b3p, 8);
                lo.c2 = 4 * take(A + r.c2p, 8); lo.c3 = 4 * take(A + r.c3p, 8);
                if (tb_first < 0) tb_first = r.tb_off;
                else if (r.tb_off != tb_last_end) { fits = false; break; }   // a phase's time-bias slices must be one contiguous piece of the row
                tb_last_end = r.tb_off + pad32(r.N);
            } else {
                const LinOpP& l = h->lin[op.p];
                lo.w1 = take(A + l.Wh, ((groups_of(l.l.K) + 1) / 2) * 128);
                lo.c2 = 4 * take(A + l.bp, 8);
            }
        }
        int n_all = n;
        if (fits && n > 0 && tail) {
            // the tail Linear's planes (2 out tiles x 2 steps) and bias join the last phase when they fit
            const LinOpP& l = h->lin[h->ops[h->fuse_hi].p];
            const unsigned KSl = (groups_of(l.l.K) + 1) / 2, need = 2 * KSl * 128 + 16;
            const unsigned tb_now = tb_first < 0 ? 0 : (unsigned)(tb_last_end - tb_first) / 4;
            if (used + need + tb_now <= (unsigned)kNarrowLdsU4) {
                NarrowLdsOp& lo = lops[n];
                memset(&lo, 0, sizeof lo);
                lo.store_out = 1;
                const float* A = h->arena;
                auto take = [&](const void* src, unsigned n_u4) -> unsigned {
                    const unsigned off = used;
                    copies.push_back(NarrowLdsCopy{src, image_base + off, n_u4});
                    used += n_u4;
                    return off;
                };
                lo.w1 = take(A + l.Wh, 2 * KSl * 128);
                lo.c2 = 4 * take(A + l.bp, 16);
                n_all = n + 1;
                if (phase_lo < n) lops[n - 1].store_out = 0;      // the run's last tensor stays in registers: only the tail reads it
            }
        }
        if (fits && n > 0) {
            close_phase(n_all);
            const size_t image_u4 = image_base;
            if (!h->nlds_ops_dev) HIPCK(hipMalloc(&h->nlds_ops_dev, (h->ops.size() + 1) * sizeof(NarrowLdsOp)));
            if (!h->nlds_copies_dev) HIPCK(hipMalloc(&h->nlds_copies_dev, (h->ops.size() + 1) * 16 * sizeof(NarrowLdsCopy)));
            if (image_u4 + 1 > h->nlds_image_cap) {
                // grow-only: the captured step graphs hold the image pointer (phase arguments are passed by value); when it has
                // to move, every graph captured with the old one goes (nothing is in flight: the stream was synchronised above)
                if (h->nlds_image) { (void)hipFree(h->nlds_image); free_graphs(h); }
                HIPCK(hipMalloc(&h->nlds_image, (image_u4 + 1) * sizeof(uint4)));
                h->nlds_image_cap = image_u4 + 1;
            }
            HIPCK(hipMemcpy(h->nlds_ops_dev, lops.data(), n_all * sizeof(NarrowLdsOp), hipMemcpyHostToDevice));
            h->nlds_tail = n_all > n;
            HIPCK(hipMemcpy(h->nlds_copies_dev, copies.data(), copies.size() * sizeof(NarrowLdsCopy), hipMemcpyHostToDevice));
            for (size_t k = 0; k < h->nlds_phases.size(); ++k) h->nlds_phases[k].image = h->nlds_image + phase_base[k];
            hipLaunchKernelGGL(k_narrow_image_build, dim3((unsigned)copies.size()), dim3(256), 0, s, (const NarrowLdsCopy*)h->nlds_copies_dev, h->nlds_image);
            h->nlds_ncopies = (int)copies.size();
            h->nlds_valid = true;
        }
    }
    return 0;
}

// returns the number of operators BEHIND the run that the launch computed as well (0, or 1: the LDS form's tail Linear)
int launch_fused(const dsg_handle* h, const RunCtx& c, hipStream_t s) {
    const int ntiles = cdiv(c.nrows, 32) * c.npass;
    const dim3 grid(cdiv(ntiles, kWavesPerBlock)), block(256);
    if (split_ctx(h, c)) {
        // small launches: the latency of one wave is the kernel time -> first-step weight planes requested a stage ahead
        const FusedOpH* tab = c.train ? h->fusedh_train_dev : h->fusedh_dev;
        if (!c.train && !c.ts && h->nlds_valid && ntiles >= h->panel_min_tiles && ntiles > h->narrow_small_max_tiles) {
            // large sampling launch: the run's planes and vectors resident in LDS, one launch per phase (dsg_split.hpp)
            const int v8nb = (h->opt_v8 && h->v8_lo >= 0) ? h->d.n_blocks : 0;
            NarrowPhases P;
            memset(&P, 0, sizeof P);
            P.n = (int)h->nlds_phases.size();
            for (int k = 0; k < P.n; ++k) P.p[k] = h->nlds_phases[k];
            const dim3 g(cdiv(ntiles, 16)), b(1024);
            const NarrowLdsOp* lops = h->nlds_ops_dev;
            if (v8nb == 2) hipLaunchKernelGGL(k_fused_narrow_lds<2>, g, b, 0, s, tab, lops, P, ntiles, c.step_ptr, h->tb_stride);
            else if (v8nb == 3) hipLaunchKernelGGL(k_fused_narrow_lds<3>, g, b, 0, s, tab, lops, P, ntiles, c.step_ptr, h->tb_stride);
            else hipLaunchKernelGGL(k_fused_narrow_lds<0>, g, b, 0, s, tab, lops, P, ntiles, c.step_ptr, h->tb_stride);
            return h->nlds_tail ? 1 : 0;
        }
        // the 8-wide bottom of the net on the vector unit (float32, dsg_narrow8.hpp) from the global image: sampling launches below the
        // LDS form's threshold, dsg_unet_forward, and the training forward (which stores what the backward pass reads)
        const bool v8 = h->opt_v8 && h->v8_lo >= 0 && h->v8_ncopies > 0 && c.cond_pre;
        const int v8nb = v8 ? h->d.n_blocks : 0, v8_at = v8 ? h->v8_lo - h->fuse_lo : -1, v8_n = v8 ? h->v8_hi - h->v8_lo : 0, nops = h->fuse_hi - h->fuse_lo;
        const int st = c.train ? 1 : 0;
        const bool pre = ntiles <= h->narrow_small_max_tiles;
#define DSG_NARROW(PRE_, NB_) hipLaunchKernelGGL((k_fused_narrow_h<PRE_, NB_>), grid, block, 0, s, tab, nops, ntiles, v8_at, v8_n, (const float*)h->v8_image, st)
        if (pre) { if (v8nb == 2) DSG_NARROW(true, 2); else if (v8nb == 3) DSG_NARROW(true, 3); else DSG_NARROW(true, 0); }
        else { if (v8nb == 2) DSG_NARROW(false, 2); else if (v8nb == 3) DSG_NARROW(false, 3); else DSG_NARROW(false, 0); }
#undef DSG_NARROW
    }
    else hipLaunchKernelGGL(k_fused_narrow, grid, block, 0, s, h->fused_dev, h->fuse_hi - h->fuse_lo, ntiles);
    return 0;
}

// wide block followed by its consuming Linear (Down/Upsample or final), both outside the fused narrow run
bool try_pair(const dsg_handle* h, int i, const RunCtx& c, hipStream_t s) {
    if (!split_ctx(h, c) || c.train || i + 1 >= (int)h->ops.size()) return false;
    const Op& a = h->ops[i];
    const Op& b = h->ops[i + 1];
    if (a.kind != OP_RES || (b.kind != OP_LIN && b.kind != OP_FINAL) || b.in0 != a.out) return false;
    const bool fuse = h->fuse_hi - h->fuse_lo >= 2;
    if (fuse && i + 1 >= h->fuse_lo && i < h->fuse_hi) return false;
    const ResP& r = h->res[a.p];
    if (r.N < 64) return false;
    if (cdiv(c.nrows, 32) * c.npass <= h->coop_max_tiles) return false;   // small launch: cooperative block, then the Linear
    BlockArgs ba; LinArgs la;
    fill_block_args(h, a, c, ba);
    fill_lin_args(h, b, c, la);
    return launch_res_lin_h(h, r, ba, h->lin[b.p], la, b.kind == OP_FINAL, h->tensors[a.out].is_skip, s);
}

// exact path: the same pairs (k_resblock_lin, dsg_kernels.hpp) when both input tensors of the block have N / 8 groups
bool try_pair_f32(const dsg_handle* h, int i, const RunCtx& c, hipStream_t s) {
    if (split_ctx(h, c) || c.train || !h->opt_f32_pair || i + 1 >= (int)h->ops.size()) return false;
    const Op& a = h->ops[i];
    const Op& b = h->ops[i + 1];
    if (a.kind != OP_RES || (b.kind != OP_LIN && b.kind != OP_FINAL) || b.in0 != a.out) return false;
    const bool fuse = h->fuse_hi - h->fuse_lo >= 2;
    if (fuse && i + 1 >= h->fuse_lo && i < h->fuse_hi) return false;
    const ResP& r = h->res[a.p];
    if (r.N < 64) return false;
    BlockArgs ba; LinArgs la;
    fill_block_args(h, a, c, ba);
    fill_lin_args(h, b, c, la);
    if (ba.in0.groups != r.N / 8 || ba.in1.groups != (r.sclin ? r.N / 8 : 0) || la.in_groups != r.N / 8) return false;
    const bool fin = b.kind == OP_FINAL;
    const int NTO = cdiv(h->lin[b.p].l.N, 32), store = h->tensors[a.out].is_skip ? 1 : 0;
    const dim3 grid(cdiv(ba.ntiles, kWavesPerBlock)), block(256);
#define DSG_TRYF(N_, SC_, NTO_, FIN_)                                                                        \
    if (r.N == N_ && r.sclin == SC_ && NTO == NTO_ && fin == FIN_) {                                         \
        hipLaunchKernelGGL((k_resblock_lin<N_, SC_, NTO_, FIN_>), grid, block, 0, s, ba, la, store);         \
        return true;                                                                                         \
    }
    DSG_TRYF(128, false, 2, false) DSG_TRYF(64, true, 4, false) DSG_TRYF(128, true, 3, true) DSG_TRYF(128, true, 1, true)
    DSG_TRYF(128, false, 1, false) DSG_TRYF(64, true, 2, false) DSG_TRYF(64, true, 1, true) DSG_TRYF(64, true, 3, true)
#undef DSG_TRYF
    return false;
}

// two consecutive down-64 blocks of a large sampling launch in one launch (k_res64_dual)
bool try_dual64(const dsg_handle* h, int i, const RunCtx& c, hipStream_t s) {
    if (!split_ctx(h, c) || c.train || i + 1 >= (int)h->ops.size()) return false;
    const Op& oa = h->ops[i];
    const Op& ob = h->ops[i + 1];
    if (oa.kind != OP_RES || ob.kind != OP_RES || ob.in0 != oa.out) return false;
    const bool fuse = h->fuse_hi - h->fuse_lo >= 2;
    if (fuse && i + 1 >= h->fuse_lo && i < h->fuse_hi) return false;
    const ResP& ra = h->res[oa.p];
    const ResP& rb = h->res[ob.p];
    if (ra.N != 64 || rb.N != 64 || ra.sclin || rb.sclin) return false;
    BlockArgs ba, bb;
    fill_block_args(h, oa, c, ba);
    fill_block_args(h, ob, c, bb);
    if (!res64_lds_ok(h, ra, ba) || !res64_lds_ok(h, rb, bb)) return false;
    if (bb.in0.wrap || ba.ntiles != bb.ntiles || ba.tiles_per_pass != bb.tiles_per_pass || ba.uncond_tiles != bb.uncond_tiles) return false;
    BlockArgsH a0, a1;
    fill_block_args_h(h, ra, ba, a0);
    fill_block_args_h(h, rb, bb, a1);
    const int ngroups = cdiv(ba.ntiles, kR64Waves);
    const dim3 grid(ngroups < h->num_cus ? ngroups : h->num_cus), block(kR64Waves * 64);
    hipLaunchKernelGGL(k_res64_dual, grid, block, 0, s, a0, a1, ngroups);
    return true;
}

// Small launches: feature_proj, then ONE launch that carries every row tile through the rest of the net (dsg_tile.hpp)
bool tile_step_ok(const dsg_handle* h, const RunCtx& c) {
    return h->opt_tile && h->tile_valid && split_ctx(h, c) && !c.train && !c.ts && h->fuse_hi - h->fuse_lo >= 2 &&
           cdiv(c.nrows, 32) * c.npass <= h->coop_max_tiles;
}
void launch_tile_step(const dsg_handle* h, const RunCtx& c, hipStream_t s) {
    launch_op(h, h->ops[0], c, s);            // feature_proj: moves the device step counter on (reverse loop), one pass when shared
    const int ntiles = cdiv(c.nrows, 32) * c.npass, nops = (int)h->ops.size();
    const bool v8 = h->opt_v8 && h->v8_lo >= 0 && h->v8_ncopies > 0 && c.cond_pre;
    const int v8nb = v8 ? h->d.n_blocks : 0, v8_at = v8 ? h->v8_lo - h->fuse_lo : -1, v8_n = v8 ? h->v8_hi - h->v8_lo : 0;
    const dim3 grid(ntiles), block(256);
#define DSG_TILE(NB_) hipLaunchKernelGGL((k_unet_tile<NB_>), grid, block, 0, s, (const FusedOpH*)h->tileops_dev, 1, nops, ntiles, h->fuse_lo, h->fuse_hi, \
                                         v8_at, v8_n, (const float*)h->v8_image)
    if (v8nb == 2) DSG_TILE(2); else if (v8nb == 3) DSG_TILE(3); else DSG_TILE(0);
#undef DSG_TILE
}

void run_unet(const dsg_handle* h, const RunCtx& c, hipStream_t s) {
    if (tile_step_ok(h, c)) { launch_tile_step(h, c, s); return; }
    const bool fuse = (!c.train || split_ctx(h, c)) && h->fuse_hi - h->fuse_lo >= 2;
    for (int i = 0; i < (int)h->ops.size(); ++i) {
        if (fuse && i == h->fuse_lo) {
            i = h->fuse_hi - 1 + launch_fused(h, c, s);
            continue;
        }
        if (try_dual64(h, i, c, s) || try_pair(h, i, c, s) || try_pair_f32(h, i, c, s)) { ++i; continue; }
        launch_op(h, h->ops[i], c, s);
    }
}

// time path for `entries` t values already in h->tvals (saves for the backward when `train`)
void run_time_path(dsg_handle* h, int entries, hipStream_t s, bool train = false) {
    const int half = h->d.proj_dim / 2, td = h->td;
    const Param* P = h->params.data();
    float *emb = nullptr, *h1pre = nullptr, *tpre = nullptr;
    float* h1s = h->h1s;
    if (train) {
        emb = h->tr_tsave; h1pre = emb + (size_t)h->tr_T * 2 * half; h1s = h1pre + (size_t)h->tr_T * td;
        tpre = h1s + (size_t)h->tr_T * td;
    }
    const dim3 grid(entries, cdiv(td, 16));
    hipLaunchKernelGGL(k_time_embed1, grid, dim3(256), 2 * half * sizeof(float), s, h->tvals, h->freq, half, P[h->temb_l1w].ptr,
                       P[h->temb_l1b].ptr, td, h1s, emb, h1pre);
    hipLaunchKernelGGL(k_time_embed2, grid, dim3(256), td * sizeof(float), s, h1s, P[h->temb_l2w].ptr, P[h->temb_l2b].ptr, td, h->st, tpre);
    const int nb = (int)h->res.size();
    hipLaunchKernelGGL(k_time_table, dim3(entries, nb * 8), dim3(256), td * sizeof(float), s, h->st, td, h->tdesc_dev, nb, h->tb, h->tb_stride);
}

// cembed[block] = Wc silu(cond * mask) for every block from the condition fragments (one launch per block)
void run_cond_embed(dsg_handle* h, int B, hipStream_t s) {
    const int tpp = cdiv(B, 32), CG = groups_of(h->d.cond_dim);
    if (h->use_split && (CG + 1) / 2 <= kCondMaxSteps) {
        // split path: every block in ONE launch
        if (h->ctile_key != h->cembed || h->ctile_cap != cap_tiles_of(h)) {
            std::vector<CondTile> tab;
            for (const ResP& r : h->res) {
                const int NG = groups_of(r.N);
                for (int lt = 0; lt < cdiv(r.N, 32); ++lt) {
                    CondTile t;
                    t.out = h->cembed + r.ce_off * (cap_tiles_of(h) / 2) + (size_t)lt * 4 * 256;
                    t.m = h->maxabs + r.ce.w;
                    t.groups = NG - 4 * lt < 4 ? NG - 4 * lt : 4;
                    t.ng_block = NG;
                    tab.push_back(t);
                }
            }
            if (!h->ctile_dev) (void)hipMalloc(&h->ctile_dev, tab.size() * sizeof(CondTile));
            (void)hipMemcpy(h->ctile_dev, tab.data(), tab.size() * sizeof(CondTile), hipMemcpyHostToDevice);
            h->ctile_n = (int)tab.size(); h->ctile_key = h->cembed; h->ctile_cap = cap_tiles_of(h);
        }
        // enough waves for ~2 per SIMD: row tiles x parts
        const int parts = tpp >= 2048 ? 1 : (tpp >= 1024 ? 2 : (tpp >= 512 ? 4 : 8));
        hipLaunchKernelGGL(k_cond_embed_h, dim3(cdiv(tpp, kWavesPerBlock), parts), dim3(256), 0, s, h->condfrag, CG,
                           reinterpret_cast<const uint4*>(h->arena + h->res[0].Wch), h->ctile_dev, h->ctile_n, tpp);
        return;
    }
    // wide blocks: one launch each; all blocks <= 32 wide: ONE launch (a wave walks the list for its tile)
    std::vector<FusedOp>& tab = h->ce_host;
    tab.clear();
    for (const ResP& r : h->res) {
        LinArgs a;
        memset(&a, 0, sizeof a);
        a.in.data = h->condfrag; a.in.groups = CG; a.in.width = h->d.cond_dim;
        a.in_width = h->d.cond_dim; a.in_groups = CG;
        a.W = h->arena + r.Wcp; a.bias = h->arena + h->zero_off;
        a.out = h->cembed + r.ce_off * (cap_tiles_of(h) / 2); a.out_stats = h->ce_stats; a.out_width = r.N;
        a.ntiles = tpp; a.tiles_per_pass = tpp; a.nrows = B;
        if (r.N > 32) { launch_lin(r.N, IN_FRAG, OUT_FRAG, false, a, s); continue; }
        FusedOp f;
        memset(&f, 0, sizeof f);
        f.kind = 1; f.N = r.N; f.l = a;
        tab.push_back(f);
    }
    if (!tab.empty()) {
        (void)hipStreamSynchronize(s);  // ce_host may still be the source of the previous upload
        (void)hipMemcpyAsync(h->ce_dev, tab.data(), tab.size() * sizeof(FusedOp), hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_fused_narrow, dim3(cdiv(tpp, kWavesPerBlock)), dim3(256), 0, s, h->ce_dev, (int)tab.size(), tpp);
    }
}

int check_bound(const dsg_handle* h) {
    if (!h) return fail("null handle");
    if (!h->bound) return fail("dsg_bind_weights has not been called");
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// training workspace + descriptor tables
// ------------------------------------------------------------------------------------------------------
int ensure_train_workspace(dsg_handle* h, int rows, int T) {
    if (ensure_workspace(h, rows, T)) return 1;
    if (h->tr_ws && rows <= h->tr_rows && T == h->tr_T) return 0;
    HIPCK(hipDeviceSynchronize());
    const int nrows = rows > h->tr_rows ? rows : h->tr_rows;
    free_train_workspace(h);
    h->tr_rows = nrows; h->tr_T = T;
    const size_t tiles = tr_tiles_of(h);
    const int D = h->d.input_dim, td = h->td, half = h->d.proj_dim / 2;
    // slab = [flat parameter gradients | dTB_b [N_b][T] for every block]
    long long so = h->total_params;
    for (auto& r : h->res) { r.dtb_off = so; so += (long long)r.N * T; }
    h->slab_stride = (size_t)((so + 63) / 64 * 64);
    const int max_chunks = 32;
    HIPCK(hipMalloc(&h->tr_ws, tiles * h->tr_per_tile * sizeof(float)));
    HIPCK(hipMemset(h->tr_ws, 0, tiles * h->tr_per_tile * sizeof(float)));
    HIPCK(hipMalloc(&h->tr_slabs, (size_t)max_chunks * h->slab_stride * sizeof(float)));
    HIPCK(hipMemset(h->tr_slabs, 0, (size_t)max_chunks * h->slab_stride * sizeof(float)));
    HIPCK(hipMalloc(&h->tr_gsum, h->slab_stride * sizeof(float)));
    // + one set per early weight-gradient part + the complete set of the step's tail (the blocks' slots AND the Linears')
    HIPCK(hipMalloc(&h->tr_gmax, (size_t)(2 + kMaxWgParts) * kMaxGmax * sizeof(unsigned)));
    h->gmax_ld = (int)tiles;
    HIPCK(hipMalloc(&h->tr_gmax_t, (size_t)kMaxGmax * h->gmax_ld * sizeof(unsigned)));
    {   // per-tile column sums of the residual blocks, one contiguous [tile][slot] region per block:
        // [dout | dh2 | dh1 | beta3 | gamma3 | beta2 | gamma2] x NP, [beta1 | gamma1] x KP
        const Param* P = h->params.data();
        size_t off = 0;
        std::vector<int4> map;
        size_t base = 0, stride = 0, slot = 0;
        auto vec = [&](int width_pad, int w0, int w1, long long d0, long long d1) {
            // padded feature order: each concat segment padded to whole groups
            const int g0 = groups_of(w0);
            for (int f = 0; f < width_pad; ++f, ++slot) {
                const int G = f >> 3, e = f & 7;
                int col = -1;
                if (G < g0) { if (8 * G + e < w0) col = 8 * G + e; }
                else { const int c = 8 * (G - g0) + e; if (c < w1) col = w0 + c; }
                map.push_back(make_int4(col >= 0 ? (int)(d0 + col) : -1, (col >= 0 && d1 >= 0) ? (int)(d1 + col) : -1, (int)(base + slot), (int)stride));
            }
        };
        for (auto& r : h->res) {
            const int NP = cdiv(r.N, 32) * 32;
            const int KGT = r.sclin ? cdiv(2 * groups_of(r.N), 4) : cdiv(r.N, 32);   // as resblock_bwd_body
            const int KP = KGT * 32;
            r.cs_off = off; r.cs_stride = (size_t)7 * NP + 2 * KP;
            base = off; stride = r.cs_stride; slot = 0;
            vec(NP, r.N, 0, P[r.l3.b].off, r.sclin ? P[r.sc.b].off : -1);
            vec(NP, r.N, 0, P[r.l2.b].off, P[r.ce.b].off);
            vec(NP, r.N, 0, P[r.l1.b].off, P[r.te.b].off);
            vec(NP, r.N, 0, P[r.n3.b].off, -1);
            vec(NP, r.N, 0, P[r.n3.w].off, -1);
            vec(NP, r.N, 0, P[r.n2.b].off, -1);
            vec(NP, r.N, 0, P[r.n2.w].off, -1);
            vec(KP, r.in0, r.in1, P[r.n1.b].off, -1);
            vec(KP, r.in0, r.in1, P[r.n1.w].off, -1);
            off += tiles * r.cs_stride;
        }
        if (off >= ((size_t)1 << 31)) { fail("training batch too large for the column-sum index (%zu floats)", off); return 1; }
        h->cs_slots = (int)map.size();
        HIPCK(hipMalloc(&h->tr_cs, off * sizeof(float)));
        HIPCK(hipMalloc(&h->cs_map_dev, map.size() * sizeof(int4)));
        HIPCK(hipMemcpy(h->cs_map_dev, map.data(), map.size() * sizeof(int4), hipMemcpyHostToDevice));
    }
    HIPCK(hipMalloc(&h->tr_ts, (size_t)nrows * sizeof(int)));
    HIPCK(hipMalloc(&h->tr_yt_rm, (size_t)nrows * D * sizeof(float)));
    HIPCK(hipMalloc(&h->tr_tsave, ((size_t)T * 2 * half + (size_t)(5 + kTimeChunks) * T * td) * sizeof(float)));
    return 0;
}

// Descriptors depend on the batch size of THIS call (tile counts, row masks) and on T: rebuilt per call (host only,
// one upload); cheap next to the step.
// Arguments of one operator's activation-gradient kernel (dsg_train_step, and the fused narrow backward's table)
void fill_res_bwd_args(const dsg_handle* h, const Op& op, int tiles, BlockBwdArgs& a, BlockBwdArgsH& ah) {
    const float* A = h->arena;
    const ResP& r = h->res[op.p];
    memset(&a, 0, sizeof a);
    a.in0 = seg_of(h, op.in0);
    if (op.in1 >= 0) a.in1 = seg_of(h, op.in1);
    a.h1 = trp(h, r.h1); a.h2 = trp(h, r.h2);
    a.gout_a = trp(h, h->tensors[op.out].ga);
    a.gout_b = h->tensors[op.out].is_skip ? trp(h, h->tensors[op.out].gb) : nullptr;
    a.W3T = A + r.W3T; a.W2T = A + r.W2T; a.W1T = A + r.W1T; a.WscT = r.sclin ? A + r.WscT : nullptr;
    a.gamma1 = A + r.g1p; a.beta1 = A + r.b1p; a.gamma2 = A + r.g2p; a.beta2 = A + r.b2p; a.gamma3 = A + r.g3p; a.beta3 = A + r.b3p;
    a.gin0 = trp(h, h->tensors[op.in0].ga);
    a.gin1 = op.in1 >= 0 ? trp(h, h->tensors[op.in1].gb) : nullptr;
    a.du1 = trp(h, r.du1); a.dh1 = trp(h, r.dh1); a.dh2 = trp(h, r.dh2);
    a.cs = h->tr_cs + r.cs_off; a.cs_stride = r.cs_stride;
    a.rs1 = trp(h, r.rs1); a.rs2 = trp(h, r.rs2); a.rs3 = trp(h, r.rs3);
    a.ntiles = tiles;
    ah.b = a;
    ah.W3Th = reinterpret_cast<const uint4*>(A + r.W3Th); ah.W2Th = reinterpret_cast<const uint4*>(A + r.W2Th);
    ah.W1Th = reinterpret_cast<const uint4*>(A + r.W1Th);
    ah.WscTh = r.sclin ? reinterpret_cast<const uint4*>(A + r.WscTh) : nullptr;
    ah.m1 = h->maxabs + r.l1.w; ah.m2 = h->maxabs + r.l2.w; ah.m3 = h->maxabs + r.l3.w;
    ah.msc = r.sclin ? h->maxabs + r.sc.w : nullptr;
    ah.gmax_t = h->tr_gmax_t; ah.gmax_ld = h->gmax_ld; ah.slot_out = r.gslot_out; ah.slot_h2 = r.gslot_h2; ah.slot_h1 = r.gslot_h1;
}
void fill_lin_bwd_args(const dsg_handle* h, const Op& op, int tiles, LinBwdArgs& a) {
    const float* A = h->arena;
    const LinOpP& l = h->lin[op.p];
    memset(&a, 0, sizeof a);
    if (op.kind == OP_FINAL) {
        a.gout_a = trp(h, h->tr_deps);
        a.gamma = A + l.gp; a.beta = A + l.betap; a.du = trp(h, l.du); a.rs = trp(h, l.rs);
    } else {
        a.gout_a = trp(h, h->tensors[op.out].ga);
        a.gout_b = h->tensors[op.out].is_skip ? trp(h, h->tensors[op.out].gb) : nullptr;
    }
    a.out_groups = groups_of(l.l.N);
    a.WT = A + l.WT;
    a.in = seg_of(h, op.in0);
    a.gin = trp(h, h->tensors[op.in0].ga);
    a.ntiles = tiles;
}

int build_train_descs(dsg_handle* h, int B, int T, hipStream_t s) {
    // Rebuilding and uploading the tables costs four pageable host-to-device copies and a stream synchronise - a pipeline
    // drain in front of every step (the enqueue of a step's ~85 launches then no longer hides behind the previous step)
    const void* key[6] = {h->tr_ws, h->ws, h->arena, h->cembed, h->condfrag, h->tb};
    if (h->td_valid && h->td_B == B && h->td_T == T && h->td_rows == h->tr_rows && h->td_split == h->use_split &&
        memcmp(key, h->td_key, sizeof key) == 0)
        return 0;
    h->td_valid = false;
    const float* A = h->arena;
    const Param* P = h->params.data();
    const int DG = groups_of(h->d.input_dim);
    const int tiles = cdiv(B, 32);
    h->tr_chunks = tiles < 32 ? tiles : 32;
    std::vector<WgradDesc> wd;
    std::vector<int> wd_op;            // operator each descriptor belongs to
    int cur_op = 0;
    // part of each operator: k for a residual block reached before fork k (only blocks: the scale of their G operands comes
    // from their own backward kernel, a plain Linear's from k_colsum at the end), -1 = the final launch on the caller's stream
    std::vector<int> op_part(h->ops.size() + 1, -1);
    h->wg_fork_ops.clear();
    // Only where the step is GPU-bound (>= 1 024 row tiles): the cross-stream hand-offs cost host time, and below that the step is
    // bound by the host's enqueue rate (measured: 32 768 rows 2.65 -> 2.53 ms, 65 536 rows 4.03 -> 3.98; 16 384 rows 2.00 -> 2.26).
    if (h->use_split && tiles >= kWgForkMinTiles) {
        int n = 0;
        size_t k = 0;
        for (int oi = (int)h->ops.size() - 1; oi >= 0 && k < h->wg_forks.size(); --oi) {
            if (h->ops[oi].kind != OP_RES) continue;
            op_part[oi + 1] = (int)k;
            if (++n == h->wg_forks[k]) { h->wg_fork_ops.push_back(oi); ++k; }
        }
        if (k < h->wg_forks.size())                   // fewer blocks than the last fork asks for: those stay in the final launch
            for (size_t i = 0; i < op_part.size(); ++i) if (op_part[i] == (int)k) op_part[i] = -1;
        // one more part: the residual blocks of the fused narrow run (17 of MSR-80c's 27 blocks, ~2 700 small units).  Their backward is
        // one launch (k_fused_narrow_bwd_h); behind it the chain still has the 64- and 128-wide down blocks to go.
        if (k == h->wg_forks.size() && h->opt_wg_narrow_part && h->fuse_hi - h->fuse_lo >= 2 && (int)h->wg_fork_ops.size() < kMaxWgParts) {
            int lowest = -1;
            for (int oi = h->fuse_hi - 1; oi >= h->fuse_lo; --oi)
                if (h->ops[oi].kind == OP_RES && op_part[oi + 1] < 0) { op_part[oi + 1] = (int)k; lowest = oi; }
            if (lowest >= 0) h->wg_fork_ops.push_back(lowest);
        }
    }
    std::vector<ColsumDesc> cd;
    std::vector<int> bwd_slots;        // slots tracked by the block backward kernels
    std::vector<const float*> gsrc;    // distinct G tensors, by first pointer: slot of max|G|
    auto slot_of = [&](const float* g) {
        for (size_t i = 0; i < gsrc.size(); ++i) if (gsrc[i] == g) return (int)i;
        gsrc.push_back(g);
        return (int)gsrc.size() - 1;
    };
    auto gseg = [&](const float* data, int width) { Seg sg; sg.data = data; sg.stats = nullptr; sg.groups = groups_of(width); sg.width = width; sg.wrap = 0; return sg; };
    auto none = [&]() { Seg sg; sg.data = nullptr; sg.stats = nullptr; sg.groups = 0; sg.width = 0; sg.wrap = 0; return sg; };
    auto wgrad = [&](const float* G0, const float* G1, int N, int amode, Seg a0, Seg a1, const float* rs, const float* gm,
                     const float* bt, long long out_off, int ld) {
        WgradDesc d;
        memset(&d, 0, sizeof d);
        d.G0 = G0; d.G1 = G1; d.N = N; d.NG = groups_of(N); d.amode = amode; d.a0 = a0; d.a1 = a1; d.rs = rs; d.gamma = gm; d.beta = bt;
        d.ts = h->tr_ts; d.onehot_n = T; d.out_off = out_off; d.ld = ld;
        d.KG = amode == A_ONEHOT ? groups_of(T) : a0.groups + a1.groups;
        d.nrows = B;
        d.gmax_slot = slot_of(G0);
        wd.push_back(d);
        wd_op.push_back(cur_op);
    };
    auto colsum = [&](const float* P0, const float* P1, int groups, Seg x0, Seg x1, const float* rs, int w0, int w1, long long o1,
                      long long o1b, long long o2) {
        ColsumDesc d;
        memset(&d, 0, sizeof d);
        d.P0 = P0; d.P1 = P1; d.groups = groups; d.x0 = x0; d.x1 = x1; d.rs = rs; d.w0 = w0; d.w1 = w1;
        d.out_off = o1; d.out_off_b = o1b; d.out2_off = o2;
        d.gmax_slot = o2 < 0 ? slot_of(P0) : -1;     // the plain sums are exactly the G tensors of the weight gradients
        cd.push_back(d);
    };
    auto grad_a = [&](int tid) { return (const float*)trp(h, h->tensors[tid].ga); };
    auto grad_b = [&](int tid) { return h->tensors[tid].is_skip ? (const float*)trp(h, h->tensors[tid].gb) : (const float*)nullptr; };

    for (const Op& op : h->ops) {
        ++cur_op;
        if (op.kind == OP_RES) {
            const ResP& r = h->res[op.p];
            const Seg in0 = seg_of(h, op.in0), in1 = op.in1 >= 0 ? seg_of(h, op.in1) : none();
            const float *ga = grad_a(op.out), *gb = grad_b(op.out);
            const float *dh1 = trp(h, r.dh1), *dh2 = trp(h, r.dh2);
            const int in = r.in0 + r.in1;
            wgrad(dh1, nullptr, r.N, A_LNSILU, in0, in1, trp(h, r.rs1), A + r.g1p, A + r.b1p, P[r.l1.w].off, in);
            if (r.sclin) wgrad(ga, gb, r.N, A_RAW, in0, in1, nullptr, nullptr, nullptr, P[r.sc.w].off, in);
            wgrad(dh2, nullptr, r.N, A_LNSILU, gseg(trp(h, r.h1), r.N), none(), trp(h, r.rs2), A + r.g2p, A + r.b2p, P[r.l2.w].off, r.N);
            wgrad(dh2, nullptr, r.N, A_RAW, gseg(h->condfrag, h->d.cond_dim), none(), nullptr, nullptr, nullptr, P[r.ce.w].off, h->d.cond_dim);
            wgrad(ga, gb, r.N, A_LNSILU, gseg(trp(h, r.h2), r.N), none(), trp(h, r.rs3), A + r.g3p, A + r.b3p, P[r.l3.w].off, r.N);
            wgrad(dh1, nullptr, r.N, A_ONEHOT, none(), none(), nullptr, nullptr, nullptr, r.dtb_off, T);
            // bias and LayerNorm gradients of the block come from the backward kernel's own column sums (k_cs_reduce); so does
            // max|G| of its three gradient tensors
            ResP& rw = h->res[op.p];
            rw.gslot_out = slot_of(ga); rw.gslot_h2 = slot_of(dh2); rw.gslot_h1 = slot_of(dh1);
            bwd_slots.push_back(rw.gslot_out); bwd_slots.push_back(rw.gslot_h2); bwd_slots.push_back(rw.gslot_h1);
        } else {
            const LinOpP& l = h->lin[op.p];
            if (op.kind == OP_FINAL) {
                const float* deps = trp(h, h->tr_deps);
                const Seg in = seg_of(h, op.in0);
                wgrad(deps, nullptr, l.l.N, A_LNSILU, in, none(), trp(h, l.rs), A + l.gp, A + l.betap, P[l.l.w].off, l.l.K);
                colsum(deps, nullptr, DG, none(), none(), nullptr, l.l.N, 0, P[l.l.b].off, -1, -1);
                colsum(trp(h, l.du), nullptr, in.groups, in, none(), trp(h, l.rs), l.l.K, 0, P[l.ln.b].off, -1, P[l.ln.w].off);
            } else {
                const float *ga = grad_a(op.out), *gb = grad_b(op.out);
                const Seg a0 = op.kind == OP_PROJ ? gseg(trp(h, h->tr_yt_frag), l.l.K) : seg_of(h, op.in0);
                wgrad(ga, gb, l.l.N, A_RAW, a0, none(), nullptr, nullptr, nullptr, P[l.l.w].off, l.l.K);
                colsum(ga, gb, groups_of(l.l.N), none(), none(), nullptr, l.l.N, 0, P[l.l.b].off, -1, -1);
            }
        }
    }
    // every G tensor of a weight gradient must be covered by a plain column sum (each Linear has a bias), or its scale is never set
    {
        std::vector<char> seen(gsrc.size(), 0);
        for (const ColsumDesc& c : cd) if (c.gmax_slot >= 0) seen[c.gmax_slot] = 1;
        for (int sl : bwd_slots) seen[sl] = 1;
        for (const WgradDesc& w : wd) if (!seen[w.gmax_slot]) { fail("internal: weight-gradient operand without a tracked maximum"); return 1; }
        if ((int)gsrc.size() > kMaxGmax) { fail("internal: too many gradient tensors (%d)", (int)gsrc.size()); return 1; }
        h->n_gmax = (int)gsrc.size();
    }
    // Launch order of the (descriptor, k-block, row-chunk) units:
    //   * operators with the longest units first (the proj_dim-wide up blocks come last in op order and would form the tail);
    //   * inside an operator, 8 row chunks at a time, every unit of the operator over those 8 chunks: workgroup i runs on XCD
    //     i % 8, so the units that re-read the same rows of dh1 / dh2 / dout / the block input are neighbours in time on the
    //     same XCD and find them in its L2.
    std::vector<WgradUnit> wu;
    {
        auto cost = [&](int di, int kb) {
            const WgradDesc& d = wd[di];
            const int nt = cdiv(d.N, 32), ntp = nt <= 1 ? 1 : (nt == 2 ? 2 : 4);
            const int ngr = d.KG - kb * 16 < 16 ? d.KG - kb * 16 : 16;
            return ntp * cdiv(ngr, 4);
        };
        struct OpUnits { int cost; std::vector<std::pair<int, int>> u; };
        std::vector<OpUnits> ou;
        for (size_t i = 0; i < wd.size(); ++i) {
            if (i == 0 || wd_op[i] != wd_op[i - 1]) ou.push_back(OpUnits{0, {}});
            for (int kb = 0; kb < cdiv(wd[i].KG, 16); ++kb) {
                ou.back().u.push_back({(int)i, kb});
                ou.back().cost = std::max(ou.back().cost, cost((int)i, kb));
            }
        }
        std::stable_sort(ou.begin(), ou.end(), [](const OpUnits& a, const OpUnits& b) { return a.cost > b.cost; });
        h->wg_part_end.clear();
        const int nparts = (int)h->wg_fork_ops.size();
        // the final part opens with its time-table (one-hot) units: once they are through, every dTB row of the step is complete and
        // the time path can run on the side stream beside the rest of the final launch (dsg_train_step)
        // ... then the other units of its residual blocks, then the plain Linears' (the scale of a block's G operands is known when
        // the block's backward kernel has run, a Linear's only after k_colsum)
        for (int part = 0; part <= nparts; ++part) {
            for (int pass = (part < nparts ? 2 : 0); pass < 3; ++pass) {
                for (const OpUnits& o : ou) {
                    const int op1 = wd_op[o.u.front().first];          // operator index + 1
                    if (op_part[op1] != (part < nparts ? part : -1)) continue;
                    const bool block = h->ops[op1 - 1].kind == OP_RES;
                    for (int c0 = 0; c0 < h->tr_chunks; c0 += 8)
                        for (const auto& u : o.u) {
                            const int cls = wd[u.first].amode == A_ONEHOT ? 0 : (block ? 1 : 2);
                            if (part == nparts && cls != pass) continue;
                            for (int c = c0; c < c0 + 8 && c < h->tr_chunks; ++c) wu.push_back(WgradUnit{u.first, u.second, c, 0});
                        }
                }
                if (part == nparts && pass == 0) h->wg_onehot_end = (int)wu.size();
                if (part == nparts && pass == 1) h->wg_blocks_end = (int)wu.size();
            }
            if (part < nparts) h->wg_part_end.push_back((int)wu.size());
        }
    }
    // fused narrow backward: the operators of the forward's narrow run, last first
    std::vector<FusedBwdOpH> fb;
    if (h->use_split && h->fuse_hi - h->fuse_lo >= 2) {
        bool ok = true;
        for (int oi = h->fuse_hi - 1; oi >= h->fuse_lo && ok; --oi) {
            const Op& op = h->ops[oi];
            FusedBwdOpH f;
            memset(&f, 0, sizeof f);
            if (op.kind == OP_RES) {
                const ResP& r = h->res[op.p];
                BlockBwdArgs a;
                f.kind = 0; f.N = r.N; f.sclin = r.sclin ? 1 : 0;
                fill_res_bwd_args(h, op, tiles, a, f.b);
                ok = r.N <= 32;
            } else if (op.kind == OP_LIN) {
                f.kind = 1; f.ot = cdiv(h->lin[op.p].l.K, 32);
                fill_lin_bwd_args(h, op, tiles, f.l);
                ok = f.ot <= 2;
            } else ok = false;
            fb.push_back(f);
        }
        if (!ok) fb.clear();
    }
    h->fbwd_n = (int)fb.size();
    if (h->fbwd_n) {
        if (!h->fbwd_dev) HIPCK(hipMalloc(&h->fbwd_dev, (h->ops.size() + 1) * sizeof(FusedBwdOpH)));
        HIPCK(hipMemcpyAsync(h->fbwd_dev, fb.data(), fb.size() * sizeof(FusedBwdOpH), hipMemcpyHostToDevice, s));
    }
    std::vector<ColsumUnit> cu;
    // chunk-major, group-minor: neighbouring waves stream neighbouring 1 KiB fragments of the same row tiles
    for (size_t i = 0; i < cd.size(); ++i)
        for (int c = 0; c < h->tr_chunks; ++c)
            for (int g = 0; g < cd[i].groups; ++g) cu.push_back(ColsumUnit{(int)i, g, c, 0});
    if (!h->wg_desc_dev) {
        HIPCK(hipMalloc(&h->wg_desc_dev, wd.size() * sizeof(WgradDesc)));
        HIPCK(hipMalloc(&h->cs_desc_dev, cd.size() * sizeof(ColsumDesc)));
    }
    if (h->wg_units != (int)wu.size() || !h->wg_unit_dev) {
        if (h->wg_unit_dev) (void)hipFree(h->wg_unit_dev);
        HIPCK(hipMalloc(&h->wg_unit_dev, wu.size() * sizeof(WgradUnit)));
    }
    if (h->cs_units != (int)cu.size() || !h->cs_unit_dev) {
        if (h->cs_unit_dev) (void)hipFree(h->cs_unit_dev);
        HIPCK(hipMalloc(&h->cs_unit_dev, cu.size() * sizeof(ColsumUnit)));
    }
    h->wg_units = (int)wu.size(); h->cs_units = (int)cu.size();
    HIPCK(hipMemcpyAsync(h->wg_desc_dev, wd.data(), wd.size() * sizeof(WgradDesc), hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(h->cs_desc_dev, cd.data(), cd.size() * sizeof(ColsumDesc), hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(h->wg_unit_dev, wu.data(), wu.size() * sizeof(WgradUnit), hipMemcpyHostToDevice, s));
    HIPCK(hipMemcpyAsync(h->cs_unit_dev, cu.data(), cu.size() * sizeof(ColsumUnit), hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));  // host vectors go out of scope
    memcpy(h->td_key, key, sizeof key);
    if (!h->r2_dev) {   // the ranges of the flat gradient that the slabs hold: everything but what the time path produces (dsg_train_step);
                        // a function of the parameter layout only (the slab stride is a multiple of 64 whatever T): built once per handle
        std::vector<std::pair<long long, long long>> skip;
        for (const ResP& r : h->res) skip.push_back({P[r.te.w].off, P[r.te.w].numel});
        for (int pi : {h->temb_l1w, h->temb_l1b, h->temb_l2w, h->temb_l2b}) skip.push_back({P[pi].off, P[pi].numel});
        std::sort(skip.begin(), skip.end());
        std::vector<long long> tab;      // starts, then prefix sums
        std::vector<long long> starts, prefix;
        long long at = 0, acc = 0;
        for (const auto& sk : skip) {
            if (sk.first > at) { starts.push_back(at); prefix.push_back(acc); acc += sk.first - at; }
            at = std::max(at, sk.first + sk.second);
        }
        if (h->total_params > at) { starts.push_back(at); prefix.push_back(acc); acc += h->total_params - at; }
        // every range on a multiple of four elements (all shipped configurations: layer widths are multiples of 4) and the slab stride
        // too: the reduce then moves 16 bytes per access, the table counts float4 units
        bool vec4 = h->slab_stride % 4 == 0;
        for (size_t i = 0; i < starts.size(); ++i) {
            const long long len = (i + 1 < starts.size() ? prefix[i + 1] : acc) - prefix[i];
            if (starts[i] % 4 || len % 4) vec4 = false;
        }
        h->r2_vec4 = vec4;
        h->r2_n = (int)starts.size(); h->r2_total = acc;
        tab = starts; tab.insert(tab.end(), prefix.begin(), prefix.end());
        if (vec4) {                                        // the same table in float4 units behind the scalar one
            for (long long v : starts) tab.push_back(v / 4);
            for (long long v : prefix) tab.push_back(v / 4);
        }
        HIPCK(hipMalloc(&h->r2_dev, tab.size() * sizeof(long long)));
        HIPCK(hipMemcpy(h->r2_dev, tab.data(), tab.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    h->td_B = B; h->td_T = T; h->td_rows = h->tr_rows; h->td_split = h->use_split; h->td_valid = true;
    return 0;
}

void small_gemm(const float* A, long long ai, long long al, const float* B, long long bl, long long bj, float* C, long long ci,
                long long cj, int M, int N, int L, int acc, hipStream_t s) {
    const long long total = (long long)M * N;
    const unsigned blocks = (unsigned)((total + 63) / 64 < 8192 ? (total + 63) / 64 : 8192);
    hipLaunchKernelGGL(k_small_gemm, dim3(blocks), dim3(256), 0, s, A, ai, al, B, bl, bj, C, ci, cj, M, N, L, acc);
}

}  // namespace

// ======================================================================================================
extern "C" {

const char* dsg_last_error(void) { return g_err.c_str(); }

#ifndef DSG_BUILD_ID_STR
#define DSG_BUILD_ID_STR "unknown"
#endif
// "DSG_BUILD_ID=<sha256 of the sources>": _lib.py compares it with the tree before loading (a stale binary must not pass tests)
const char* dsg_build_id(void) {
    static const char id[] = "DSG_BUILD_ID=" DSG_BUILD_ID_STR;
    return id + 13;
}

dsg_handle* dsg_create(const dsg_unet_desc* desc) {
    if (!desc) { fail("null desc"); return nullptr; }
    const dsg_unet_desc d = *desc;
    if (d.n_res < 1 || d.n_res > 8 || d.n_blocks < 1 || d.input_dim < 1 || d.cond_dim < 1) {
        fail("bad UNet1D descriptor"); return nullptr;
    }
    if (d.input_dim > 128 || d.cond_dim > 4096) { fail("input_dim > 128 is not supported"); return nullptr; }
    if (!width_supported(d.proj_dim) || d.proj_dim < 8) {
        fail("proj_dim %d unsupported (supported block widths: 4 (dims only), 8, 16, 32, 64, 128)", d.proj_dim); return nullptr;
    }
    for (int i = 0; i < d.n_res; ++i)
        if (!width_supported(d.dims[i])) { fail("dims[%d]=%d unsupported (4, 8, 16, 32, 64, 128)", i, d.dims[i]); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fail("no HIP device: libdiffsg_hip needs an MI355X"); return nullptr; }

    dsg_handle* h = new dsg_handle();
    h->d = d;
    h->td = 4 * d.proj_dim;
    (void)hipGetDevice(&h->device);
    // ---- parameter table + plan, in the registration order of UNet1D.__init__ (UNetCF.py:272-316)
    LinOpP proj;
    proj.l = add_linear(h, "feature_proj", d.input_dim, d.proj_dim);
    h->lin.push_back(proj);
    {
        LinearP t1 = add_linear(h, "time_emb.lin1", d.proj_dim, h->td);
        LinearP t2 = add_linear(h, "time_emb.lin2", h->td, h->td);
        h->temb_l1w = t1.w; h->temb_l1b = t1.b; h->temb_l2w = t2.w; h->temb_l2b = t2.b;
    }
    std::vector<int> skips;
    int cur = add_tensor(h, d.proj_dim, true);
    h->ops.push_back(Op{OP_PROJ, 0, -1, -1, cur, "feature_proj"});
    skips.push_back(cur);
    int w = d.proj_dim, idx = 0;
    char nm[64];
    auto push_down_res = [&](int width) {
        snprintf(nm, sizeof nm, "down.%d.res", idx);
        const int p = add_res(h, nm, width, 0, width);
        const int out = add_tensor(h, width, true);
        h->ops.push_back(Op{OP_RES, p, cur, -1, out, nm});
        cur = out; skips.push_back(cur); ++idx;
    };
    for (int i = 0; i < d.n_res; ++i) {
        for (int b = 0; b < d.n_blocks; ++b) push_down_res(w);
        snprintf(nm, sizeof nm, "down.%d.lin", idx);
        LinOpP l; l.l = add_linear(h, nm, w, d.dims[i]);
        h->lin.push_back(l);
        const int out = add_tensor(h, d.dims[i], true);
        h->ops.push_back(Op{OP_LIN, (int)h->lin.size() - 1, cur, -1, out, nm});
        if (i == d.n_res - 1 && w == 16 && d.dims[i] == 8 && (d.n_blocks == 2 || d.n_blocks == 3)) h->v8_lo = (int)h->ops.size() - 1;
        cur = out; skips.push_back(cur); ++idx;
        w = d.dims[i];
        if (i == d.n_res - 1)
            for (int b = 0; b < d.n_blocks; ++b) push_down_res(w);
    }
    for (int m = 1; m <= 2; ++m) {
        snprintf(nm, sizeof nm, "middle.res%d", m);
        const int p = add_res(h, nm, w, 0, w);
        const int out = add_tensor(h, w, false);
        h->ops.push_back(Op{OP_RES, p, cur, -1, out, nm});
        cur = out;
    }
    idx = 0;
    auto push_up_res = [&](int width) {
        snprintf(nm, sizeof nm, "up.%d.res", idx);
        const int sk = skips.back(); skips.pop_back();
        const int p = add_res(h, nm, width, h->tensors[sk].width, width);
        const int out = add_tensor(h, width, false);
        h->ops.push_back(Op{OP_RES, p, cur, sk, out, nm});
        cur = out; ++idx;
    };
    for (int i = d.n_res - 1; i >= 0; --i) {
        for (int b = 0; b < d.n_blocks + 1; ++b) push_up_res(w);
        const int nw = i > 0 ? d.dims[i - 1] : d.proj_dim;
        snprintf(nm, sizeof nm, "up.%d.lin", idx);
        LinOpP l; l.l = add_linear(h, nm, w, nw);
        h->lin.push_back(l);
        const int out = add_tensor(h, nw, false);
        h->ops.push_back(Op{OP_LIN, (int)h->lin.size() - 1, cur, -1, out, nm});
        if (i == d.n_res - 1 && h->v8_lo >= 0 && nw == 16) h->v8_hi = (int)h->ops.size();
        cur = out; ++idx;
        w = nw;
        if (i == 0)
            for (int b = 0; b < d.n_blocks + 1; ++b) push_up_res(w);
    }
    {
        LinOpP f;
        f.ln = add_norm(h, "norm", w);
        f.l = add_linear(h, "final", w, d.input_dim);
        f.lnact = true;
        h->lin.push_back(f);
        h->ops.push_back(Op{OP_FINAL, (int)h->lin.size() - 1, cur, -1, -1, "final"});
    }
    for (const ResP& r : h->res)
        if ((r.in1 && r.in1 != r.in0) || (r.sclin != (r.in1 != 0))) { fail("internal: unexpected block shape"); delete h; return nullptr; }
    carve(h);
    {   // longest consecutive run of narrow operators
        int best_lo = 0, best_hi = 0, lo = -1;
        for (int i = 0; i <= (int)h->ops.size(); ++i) {
            const bool f = i < (int)h->ops.size() && fusable(h, h->ops[i]);
            if (f && lo < 0) lo = i;
            if (!f && lo >= 0) { if (i - lo > best_hi - best_lo) { best_lo = lo; best_hi = i; } lo = -1; }
        }
        h->fuse_lo = best_lo; h->fuse_hi = best_hi;
        // the float32 section must be the shape dsg_narrow8.hpp is written for and lie inside the fused run
        if (h->v8_lo < 0 || h->v8_hi != h->v8_lo + 2 * d.n_blocks + 5 || h->v8_lo < h->fuse_lo || h->v8_hi > h->fuse_hi) h->v8_lo = h->v8_hi = -1;
        for (int i = h->v8_lo + 1; h->v8_lo >= 0 && i + 1 < h->v8_hi; ++i) {
            const Op& o = h->ops[i];
            const bool up = i - h->v8_lo > d.n_blocks + 2;
            if (o.kind != OP_RES || h->res[o.p].N != 8 || h->res[o.p].in0 != 8 || h->res[o.p].in1 != (up ? 8 : 0) || o.in0 != h->ops[i - 1].out ||
                h->res[o.p].tb_off != h->res[h->ops[h->v8_lo + 1].p].tb_off + 32 * (i - h->v8_lo - 1) || o.p != h->ops[h->v8_lo + 1].p + (i - h->v8_lo - 1) ||
                (up && o.in1 != h->ops[h->v8_lo + d.n_blocks - (i - h->v8_lo - d.n_blocks - 3)].out))
                h->v8_lo = h->v8_hi = -1;
        }
    }
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            h->num_cus = cus;
    }
    bool ok = hipMalloc(&h->ce_dev, (h->res.size() + 1) * sizeof(FusedOp)) == hipSuccess &&
              hipMalloc(&h->fusedh_dev, (h->ops.size() + 1) * sizeof(FusedOpH)) == hipSuccess &&
              hipMalloc(&h->tileops_dev, (h->ops.size() + 1) * sizeof(FusedOpH)) == hipSuccess &&
              hipMalloc(&h->maxabs, (h->params.size() + 1) * sizeof(float)) == hipSuccess &&
              hipMalloc(&h->fused_dev, (h->ops.size() + 1) * sizeof(FusedOp)) == hipSuccess &&
              hipMalloc(&h->arena, h->arena_floats * sizeof(float)) == hipSuccess &&
              hipMemset(h->arena, 0, h->arena_floats * sizeof(float)) == hipSuccess &&
              hipMalloc(&h->tdesc_dev, h->res.size() * sizeof(TimeBlockDesc)) == hipSuccess &&
              hipMalloc(&h->freq, (d.proj_dim / 2) * sizeof(float)) == hipSuccess &&
              hipMalloc(&h->red, 2 * kRedBlocks * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->step_dev, 128) == hipSuccess &&
              hipMemset(h->step_dev, 0, 128) == hipSuccess &&
              hipMalloc(&h->call_dev, sizeof(CallParams)) == hipSuccess &&
              hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
    if (ok) h->range_flag = h->step_dev + 16;      // its own 64-byte line of the same small allocation
    if (ok) {
        std::vector<OpConstDesc> oc;
        for (const ResP& r : h->res) oc.push_back(OpConstDesc{r.l1.w, r.l2.w, r.l3.w, r.sclin ? r.sc.w : -1});
        for (const LinOpP& l : h->lin) oc.push_back(OpConstDesc{l.l.w, -1, -1, -1});
        ok = hipMalloc(&h->opc_desc_dev, oc.size() * sizeof(OpConstDesc)) == hipSuccess &&
             hipMalloc(&h->opc_dev, oc.size() * 4 * sizeof(float)) == hipSuccess &&
             hipMemcpy(h->opc_desc_dev, oc.data(), oc.size() * sizeof(OpConstDesc), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        // freq[k] = exp(k * -(ln 1e4 / (half-1))) in float32 (UNetCF.py:37-38)
        const int half = d.proj_dim / 2;
        std::vector<float> f(half);
        const float c = (float)(-(log(10000.0) / (half - 1)));
        for (int k = 0; k < half; ++k) f[k] = expf((float)k * c);
        ok = hipMemcpy(h->freq, f.data(), half * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { fail("device allocation failed in dsg_create"); dsg_destroy(h); return nullptr; }
    return h;
}

void dsg_destroy(dsg_handle* h) {
    if (h && h->nlds_ops_dev) (void)hipFree(h->nlds_ops_dev);
    if (h && h->nlds_copies_dev) (void)hipFree(h->nlds_copies_dev);
    if (h && h->nlds_image) (void)hipFree(h->nlds_image);
    if (h && h->v8_image) (void)hipFree(h->v8_image);
    if (h && h->v8_copies_dev) (void)hipFree(h->v8_copies_dev);
    if (h && h->seeds_dev) (void)hipFree(h->seeds_dev);
    if (h && h->red_chunks) (void)hipFree(h->red_chunks);
    if (h)
        for (auto& e : h->tev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (!h) return;
    (void)hipDeviceSynchronize();
    free_workspace(h);
    void* ptrs[] = {h->arena, h->tdesc_dev, h->pack_dev, h->freq, h->red, h->step_dev, h->call_dev, h->fused_dev, h->maxabs,
                    (void*)h->mx_ptrs_dev, h->mx_numel_dev, h->mx_idx_dev, h->packh_dev, h->opc_desc_dev, h->opc_dev, h->fusedh_dev, h->tileops_dev, h->fusedh_train_dev, h->tw_dst_dev, (void*)h->tw_src_dev, h->r2_dev, h->ce_dev, h->ctile_dev};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->fbwd_dev) (void)hipFree(h->fbwd_dev);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    for (auto& e : h->ev_tail) if (e) (void)hipEventDestroy(e);
    if (h->range_pinned) (void)hipHostFree(h->range_pinned);
    delete h;
}

int dsg_param_count(const dsg_handle* h) { return h ? (int)h->params.size() : 0; }
const char* dsg_param_name(const dsg_handle* h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : "";
}
long long dsg_param_numel(const dsg_handle* h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].numel : -1;
}
long long dsg_param_total(const dsg_handle* h) { return h ? h->total_params : 0; }

int dsg_bind_weights(dsg_handle* h, const float* const* ptrs, int n, void* stream) {
    if (!h) return fail("null handle");
    if (n != (int)h->params.size()) return fail("dsg_bind_weights: got %d pointers, the model has %d tensors", n, (int)h->params.size());
    for (int i = 0; i < n; ++i)
        if (!ptrs[i]) return fail("dsg_bind_weights: null pointer for %s", h->params[i].name.c_str());
    hipStream_t s = (hipStream_t)stream;
    const bool same = h->bound && h->bound_ptrs.size() == (size_t)n && memcmp(h->bound_ptrs.data(), ptrs, n * sizeof(float*)) == 0;
    if (!same) {
        // (re)build the pack descriptor table: ONE grouped launch re-packs everything after each optimizer step.
        // The tables below are rewritten with synchronous copies: kernels of an earlier call on `s` (a non-blocking stream
        // does not order with them) may still be reading the old ones
        HIPCK(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) h->params[i].ptr = ptrs[i];
        h->bound_ptrs.assign(ptrs, ptrs + n);
        if (h->v8_lo >= 0) {
            // copy list of the float32 section's image: raw matrices and vectors in the order of V8SecL / V8BlockL (dsg_narrow8.hpp)
            const unsigned sec_u4 = (unsigned)(h->d.n_blocks == 3 ? V8SecL<3>::SIZE : V8SecL<2>::SIZE) / 4;
            std::vector<NarrowLdsCopy> cp;
            unsigned used = 0;
            auto take = [&](const void* src, unsigned n_u4) { cp.push_back(NarrowLdsCopy{src, used, n_u4}); used += n_u4; };
            const Param* P = h->params.data();
            const float* A = h->arena;
            const LinOpP& ld = h->lin[h->ops[h->v8_lo].p];
            take(P[ld.l.w].ptr, 32); take(P[ld.l.b].ptr, 2);
            for (int k = h->v8_lo + 1; k + 1 < h->v8_hi; ++k) {
                const ResP& r = h->res[h->ops[k].p];
                const unsigned k1 = r.sclin ? 16 : 8;
                take(P[r.l1.w].ptr, 2 * k1); take(P[r.l2.w].ptr, 16); take(P[r.l3.w].ptr, 16);
                if (r.sclin) take(P[r.sc.w].ptr, 32);
                take(P[r.n1.w].ptr, k1 / 4); take(P[r.n1.b].ptr, k1 / 4);
                take(P[r.n2.w].ptr, 2); take(P[r.n2.b].ptr, 2); take(P[r.n3.w].ptr, 2); take(P[r.n3.b].ptr, 2);
                take(A + r.c2p, 2); take(A + r.c3p, 2);
            }
            const LinOpP& lu = h->lin[h->ops[h->v8_hi - 1].p];
            take(P[lu.l.w].ptr, 32); take(P[lu.l.b].ptr, 4);
            if (used != sec_u4) return fail("internal: float32 section image is %u uint4, expected %u", used, sec_u4);
            if (!h->v8_image) {
                HIPCK(hipMalloc(&h->v8_image, (size_t)sec_u4 * sizeof(uint4)));
                HIPCK(hipMalloc(&h->v8_copies_dev, cp.size() * sizeof(NarrowLdsCopy)));
            }
            HIPCK(hipMemcpy(h->v8_copies_dev, cp.data(), cp.size() * sizeof(NarrowLdsCopy), hipMemcpyHostToDevice));
            h->v8_ncopies = (int)cp.size();
        }
        const Param* P = h->params.data();
        float* A = h->arena;
        std::vector<PackDesc> pd;
        long long blk = 0;
        auto push = [&](int kind, const float* a, const float* b, size_t off, int N, int Ktot, int w0, int w1, int T, long long total) {
            PackDesc d;
            d.a = a; d.b = b; d.dst = A + off; d.kind = kind; d.N = N; d.Ktot = Ktot; d.w0 = w0; d.w1 = w1; d.T = T; d.total = total;
            d.blk_begin = blk;
            blk += (total + kPackPerBlock - 1) / kPackPerBlock;
            pd.push_back(d);
        };
        auto pack = [&](const LinearP& l, int w0, int w1, size_t off) {
            const int NT = cdiv(l.N, 32);
            push(0, P[l.w].ptr, nullptr, off, l.N, l.K, w0, w1, NT, (long long)NT * (groups_of(w0) + groups_of(w1)) * 256);
        };
        auto packT = [&](const LinearP& l, int w0, int w1, size_t off) {
            const int OT = cdiv(groups_of(w0) + groups_of(w1), 4);
            push(1, P[l.w].ptr, nullptr, off, l.N, l.K, w0, w1, OT, (long long)OT * groups_of(l.N) * 256);
        };
        auto padv = [&](const float* a, const float* b, int w0, int w1, size_t off, int npad) { push(2, a, b, off, 0, 0, w0, w1, 0, npad); };
        std::vector<TimeBlockDesc> td(h->res.size());
        for (size_t i = 0; i < h->res.size(); ++i) {
            const ResP& r = h->res[i];
            const int NT = cdiv(r.N, 32), KG = groups_of(r.in0) + groups_of(r.in1);
            pack(r.l1, r.in0, r.in1, r.W1p);
            padv(P[r.n1.w].ptr, nullptr, r.in0, r.in1, r.g1p, KG * 8 + 32);
            padv(P[r.n1.b].ptr, nullptr, r.in0, r.in1, r.b1p, KG * 8 + 32);
            pack(r.l2, r.N, 0, r.W2p);
            padv(P[r.n2.w].ptr, nullptr, r.N, 0, r.g2p, NT * 32);
            padv(P[r.n2.b].ptr, nullptr, r.N, 0, r.b2p, NT * 32);
            padv(P[r.l2.b].ptr, P[r.ce.b].ptr, r.N, 0, r.c2p, NT * 32);
            pack(r.ce, h->d.cond_dim, 0, r.Wcp);
            pack(r.l3, r.N, 0, r.W3p);
            padv(P[r.n3.w].ptr, nullptr, r.N, 0, r.g3p, NT * 32);
            padv(P[r.n3.b].ptr, nullptr, r.N, 0, r.b3p, NT * 32);
            padv(P[r.l3.b].ptr, r.sclin ? P[r.sc.b].ptr : nullptr, r.N, 0, r.c3p, NT * 32);
            if (r.sclin) { pack(r.sc, r.in0, r.in1, r.Wscp); packT(r.sc, r.in0, r.in1, r.WscT); }
            packT(r.l1, r.in0, r.in1, r.W1T);
            packT(r.l2, r.N, 0, r.W2T);
            packT(r.l3, r.N, 0, r.W3T);
            td[i] = TimeBlockDesc{P[r.te.w].ptr, P[r.te.b].ptr, P[r.l1.b].ptr, r.N, r.tb_off};
        }
        for (const LinOpP& l : h->lin) {
            const int NT = cdiv(l.l.N, 32), KG = groups_of(l.l.K);
            pack(l.l, l.l.K, 0, l.Wp);
            packT(l.l, l.l.K, 0, l.WT);
            padv(P[l.l.b].ptr, nullptr, l.l.N, 0, l.bp, NT * 32);
            if (l.lnact) {
                padv(P[l.ln.w].ptr, nullptr, l.l.K, 0, l.gp, KG * 8 + 32);
                padv(P[l.ln.b].ptr, nullptr, l.l.K, 0, l.betap, KG * 8 + 32);
            }
        }
        {   // fp16-split planes of the wide blocks
            std::vector<const float*> mp; std::vector<long long> mn; std::vector<PackHDesc> hd;
            h->mx_param.clear();
            long long hb = 0;
            auto want = [&](const LinearP& l) {
                for (size_t i = 0; i < h->mx_param.size(); ++i) if (h->mx_param[i] == l.w) return;
                h->mx_param.push_back(l.w); mp.push_back(P[l.w].ptr); mn.push_back(P[l.w].numel);
            };
            auto pushh = [&](const LinearP& l, const LinearP* pair, int role, int w0, int w1, size_t off) {
                PackHDesc d;
                const int NT = cdiv(l.N, 32), KS = (groups_of(w0) + 1) / 2 + (groups_of(w1) + 1) / 2;
                d.W = P[l.w].ptr; d.dst = reinterpret_cast<uint4*>(A + off); d.m_self = h->maxabs + l.w; d.m_pair = pair ? h->maxabs + pair->w : nullptr;
                d.role = role; d.N = l.N; d.Ktot = l.K; d.w0 = w0; d.w1 = w1; d.NT = NT; d.total = (long long)NT * KS * 128; d.blk_begin = hb;
                d.transposed = 0;
                hb += (d.total + 255) / 256;
                hd.push_back(d);
            };
            auto pushhT = [&](const LinearP& l, const LinearP* pair, int role, int w0, int w1, size_t off) {
                PackHDesc d;
                const int OT = cdiv(groups_of(w0) + groups_of(w1), 4), KS = (groups_of(l.N) + 1) / 2;
                d.W = P[l.w].ptr; d.dst = reinterpret_cast<uint4*>(A + off); d.m_self = h->maxabs + l.w; d.m_pair = pair ? h->maxabs + pair->w : nullptr;
                d.role = role; d.N = l.N; d.Ktot = l.K; d.w0 = w0; d.w1 = w1; d.NT = OT; d.total = (long long)OT * KS * 128; d.blk_begin = hb;
                d.transposed = 1;
                hb += (d.total + 255) / 256;
                hd.push_back(d);
            };
            for (const ResP& r : h->res) {
                if (!r.split) continue;
                want(r.l1); want(r.l2); want(r.l3); want(r.ce);
                pushh(r.ce, nullptr, 0, h->d.cond_dim, 0, r.Wch);
                if (r.sclin) want(r.sc);
                pushh(r.l1, nullptr, 0, r.in0, r.in1, r.W1h);
                pushh(r.l2, nullptr, 0, r.N, 0, r.W2h);
                if (r.sclin) { pushh(r.l3, &r.sc, 1, r.N, 0, r.W3h); pushh(r.sc, &r.l3, 2, r.in0, r.in1, r.Wsch); }
                else pushh(r.l3, nullptr, 0, r.N, 0, r.W3h);
                pushhT(r.l1, nullptr, 0, r.in0, r.in1, r.W1Th);
                pushhT(r.l2, nullptr, 0, r.N, 0, r.W2Th);
                if (r.sclin) { pushhT(r.l3, &r.sc, 1, r.N, 0, r.W3Th); pushhT(r.sc, &r.l3, 2, r.in0, r.in1, r.WscTh); }
                else pushhT(r.l3, nullptr, 0, r.N, 0, r.W3Th);
            }
            for (const LinOpP& l : h->lin) { want(l.l); pushh(l.l, nullptr, 0, l.l.K, 0, l.Wh); }
            h->mx_n = (int)mp.size(); h->packh_n = (int)hd.size(); h->packh_blocks = hb;
            if (h->mx_n) {
                if (!h->mx_ptrs_dev) {
                    HIPCK(hipMalloc(&h->mx_ptrs_dev, mp.size() * sizeof(float*)));
                    HIPCK(hipMalloc(&h->mx_numel_dev, mn.size() * sizeof(long long)));
                    HIPCK(hipMalloc(&h->packh_dev, hd.size() * sizeof(PackHDesc)));
                    HIPCK(hipMalloc(&h->mx_idx_dev, mp.size() * sizeof(int)));
                }
                HIPCK(hipMemcpy(h->mx_idx_dev, h->mx_param.data(), mp.size() * sizeof(int), hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(h->mx_ptrs_dev, mp.data(), mp.size() * sizeof(float*), hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(h->mx_numel_dev, mn.data(), mn.size() * sizeof(long long), hipMemcpyHostToDevice));
                HIPCK(hipMemcpy(h->packh_dev, hd.data(), hd.size() * sizeof(PackHDesc), hipMemcpyHostToDevice));
            }
        }
        {   // time_emb.weight row tables (training time-path backward)
            std::vector<long long> dst; std::vector<const float*> src;
            for (const ResP& r : h->res)
                for (int n = 0; n < r.N; ++n) { dst.push_back(P[r.te.w].off + (long long)n * h->td); src.push_back(P[r.te.w].ptr + (size_t)n * h->td); }
            h->tw_rows = (int)dst.size();
            if (!h->tw_dst_dev) {
                HIPCK(hipMalloc(&h->tw_dst_dev, dst.size() * sizeof(long long)));
                HIPCK(hipMalloc(&h->tw_src_dev, src.size() * sizeof(float*)));
            }
            HIPCK(hipMemcpy(h->tw_dst_dev, dst.data(), dst.size() * sizeof(long long), hipMemcpyHostToDevice));
            HIPCK(hipMemcpy(h->tw_src_dev, src.data(), src.size() * sizeof(float*), hipMemcpyHostToDevice));
        }
        if (!h->pack_dev) HIPCK(hipMalloc(&h->pack_dev, pd.size() * sizeof(PackDesc)));
        h->pack_n = (int)pd.size();
        h->pack_blocks = blk;
        HIPCK(hipMemcpyAsync(h->pack_dev, pd.data(), pd.size() * sizeof(PackDesc), hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(h->tdesc_dev, td.data(), td.size() * sizeof(TimeBlockDesc), hipMemcpyHostToDevice, s));
        HIPCK(hipStreamSynchronize(s));  // pd / td are host temporaries
    }
    // (k_pack_grouped also clears the max|W| words for k_maxabs; its grid of >= params / 256 blocks covers them)
    if ((long long)h->pack_blocks * 256 < (long long)h->params.size()) return fail("internal: pack grid smaller than the parameter table");
    hipLaunchKernelGGL(k_pack_grouped, dim3((unsigned)h->pack_blocks), dim3(256), 0, s, h->pack_dev, h->pack_n, h->maxabs, (int)h->params.size());
    if (h->mx_n) {
        // max|W| of every split-packed weight -> maxabs[param index]; the pack and the block kernels derive the same
        // power-of-two scale from it on the device (no host round trip); k_pack_h also writes the operators' un-scale constants
        hipLaunchKernelGGL(k_maxabs, dim3(h->mx_n, kMaxabsSlices), dim3(256), 0, s, h->mx_ptrs_dev, h->mx_numel_dev, h->mx_idx_dev, h->maxabs);
        const int nopc = (int)(h->res.size() + h->lin.size());
        if ((long long)h->packh_blocks * 256 < nopc) return fail("internal: plane-pack grid smaller than the operator table");
        hipLaunchKernelGGL(k_pack_h, dim3((unsigned)h->packh_blocks), dim3(256), 0, s, h->packh_dev, h->packh_n, (const float*)h->maxabs,
                           (const OpConstDesc*)h->opc_desc_dev, nopc, h->opc_dev);
    }
    // the LDS image of the narrow run (k_fused_narrow_lds) is a COPY of arena pieces: re-gather it from the planes packed above,
    // into the same buffer (cached graphs keep its pointer) -- otherwise sample() after an optimizer step, load_state_dict or an
    // EMA swap would run the narrow run on the previous weights while every other block uses the new ones
    if (h->v8_ncopies > 0)      // first: the LDS image below takes the section from this buffer
        hipLaunchKernelGGL(k_narrow_image_build, dim3((unsigned)h->v8_ncopies), dim3(64), 0, s, (const NarrowLdsCopy*)h->v8_copies_dev,
                           reinterpret_cast<uint4*>(h->v8_image));
    if (h->nlds_image && h->nlds_ncopies > 0)
        hipLaunchKernelGGL(k_narrow_image_build, dim3((unsigned)h->nlds_ncopies), dim3(256), 0, s, (const NarrowLdsCopy*)h->nlds_copies_dev, h->nlds_image);
    HIPCK(hipGetLastError());
    h->bound = true;
    return 0;
}

// The settings / status entry points synchronise and free graphs on the HANDLE's device, whatever device is current in the calling
// thread (a model on cuda:1 queried while cuda:0 was current synchronised the wrong device and could read the range
// flag before the sampling kernels had finished).
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const dsg_handle* h) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != h->device) (void)hipSetDevice(h->device); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int dsg_set_precision(dsg_handle* h, int mode) {
    if (!h) return fail("null handle");
    DeviceGuard dg(h);
    if (mode != DSG_PRECISION_SPLIT_F16 && mode != DSG_PRECISION_F32_MFMA) return fail("unknown precision mode %d", mode);
    const bool split = mode == DSG_PRECISION_SPLIT_F16;
    if (split != h->use_split) { (void)hipDeviceSynchronize(); free_graphs(h); h->use_split = split; }
    return 0;
}

int dsg_set_renorm_hook(dsg_handle* h, double* stats3, void (*reduce)(void*), void* user) {
    if (!h) return fail("null handle");
    if (reduce && !stats3) return fail("dsg_set_renorm_hook: a reduce function needs the 3-double device buffer");
    // the cached step graphs never contain the hook branch (enqueue_step is captured with use_hook = false and neither the
    // callback nor `stats3` is baked into a node), so they stay valid across installs and removals: no synchronise, no re-capture
    h->renorm_stats = stats3; h->renorm_fn = reduce; h->renorm_user = user;
    return 0;
}

int dsg_set_option(dsg_handle* h, int option, int value) {
    if (!h) return fail("null handle");
    DeviceGuard dg(h);
    switch (option) {
        case DSG_OPT_NARROW_VALU8:
            if ((value != 0) != h->opt_v8) {
                (void)hipDeviceSynchronize();
                free_graphs(h);                 // the captured narrow-run launches hold the plan made under the other setting
                h->fused_sig.valid = false;
                h->opt_v8 = value != 0;
            }
            return 0;
        case DSG_OPT_TRAIN_TIME_BESIDE: h->opt_time_beside = value != 0; return 0;
        case DSG_OPT_PANEL_HALF:
            if ((value != 0) != h->opt_panel_half) { (void)hipDeviceSynchronize(); free_graphs(h); h->opt_panel_half = value != 0; }
            return 0;
        case DSG_OPT_F32_PAIR:
            if ((value != 0) != h->opt_f32_pair) { (void)hipDeviceSynchronize(); free_graphs(h); h->opt_f32_pair = value != 0; }
            return 0;
        case DSG_OPT_TILE_STEP:
            if ((value != 0) != h->opt_tile) { (void)hipDeviceSynchronize(); free_graphs(h); h->opt_tile = value != 0; }
            return 0;
        case DSG_OPT_WGRAD_NARROW_PART:
            if ((value != 0) != h->opt_wg_narrow_part) { (void)hipDeviceSynchronize(); h->opt_wg_narrow_part = value != 0; h->td_valid = false; }
            return 0;
        default: return fail("dsg_set_option: unknown option %d", option);
    }
}

int dsg_range_status(dsg_handle* h, int* exceeded) {
    if (!h || !exceeded) return fail("dsg_range_status: null argument");
    DeviceGuard dg(h);
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipMemcpy(exceeded, h->range_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (*exceeded) HIPCK(hipMemset(h->range_flag, 0, sizeof(int)));
    return 0;
}

int dsg_range_status_stream(dsg_handle* h, int* exceeded, void* stream) {
    if (!h || !exceeded) return fail("dsg_range_status_stream: null argument");
    DeviceGuard dg(h);
    hipStream_t s = (hipStream_t)stream;
    // Read AND clear in one step, on `s`: a one-thread kernel exchanges the flag with 0 and writes what it found into a pinned word (ADVICE r5:
    // "copy, then memset" lost a flag raised between the two by kernels of another stream -- twin handles, the side stream).  The pinned word
    // belongs to the handle (freed in dsg_destroy); the copy is truly asynchronous and only `s` is waited for.
    if (!h->range_pinned) HIPCK(hipHostMalloc(reinterpret_cast<void**>(&h->range_pinned), sizeof(int), hipHostMallocDefault));
    hipLaunchKernelGGL(k_flag_take, dim3(1), dim3(64), 0, s, h->range_flag, h->range_pinned);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(s));
    *exceeded = *h->range_pinned;
    return 0;
}

int dsg_set_launch_policy(dsg_handle* h, int coop_max_tiles, int narrow_small_max_tiles) {
    if (!h) return fail("null handle");
    DeviceGuard dg(h);
    const int c = coop_max_tiles < 0 ? kCoopMaxTilesDefault : coop_max_tiles;
    const int n = narrow_small_max_tiles < 0 ? kNarrowSmallMaxTilesDefault : narrow_small_max_tiles;
    if (c != h->coop_max_tiles || n != h->narrow_small_max_tiles) {
        (void)hipDeviceSynchronize();
        free_graphs(h);                 // the captured step graphs hold the kernel forms chosen at capture time
        h->coop_max_tiles = c; h->narrow_small_max_tiles = n;
        h->panel_min_tiles = c == 0 ? 0 : kPanelMinTilesDefault;
    }
    return 0;
}

int dsg_reserve(dsg_handle* h, int max_rows, int max_entries) {
    if (!h) return fail("null handle");
    return ensure_workspace(h, max_rows, max_entries);
}

int dsg_unet_forward(dsg_handle* h, const float* x, const float* t, const float* cond, const float* cond_mask, float* out,
                     int B, void* stream) {
    if (check_bound(h)) return 1;
    if (B < 1) return fail("B must be >= 1");
    if (ensure_workspace(h, B, B)) return 1;
    hipStream_t s = (hipStream_t)stream;
    HIPCK(hipMemcpyAsync(h->tvals, t, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
    run_time_path(h, B, s);
    const int tpp = cdiv(B, 32), CG = groups_of(h->d.cond_dim);
    hipLaunchKernelGGL(k_cond_frag, dim3(cdiv(tpp * CG * 256, 256)), dim3(256), 0, s, cond, cond_mask, B, h->d.cond_dim, CG,
                       h->condfrag, tpp);
    // the handle's precision mode applies here as in dsg_sample / dsg_train_step: the split kernels take the condition embedding
    // Wc silu(cond * mask) precomputed (a masked row contributes exactly 0), the exact-f32 kernels compute it in the block
    if (h->use_split) run_cond_embed(h, B, s);
    RunCtx c{B, 1, 0, x, out, nullptr, h->ts_ident, false, h->use_split};
    if (prepare_fused(h, c, s)) return 1;
    run_unet(h, c, s);
    HIPCK(hipGetLastError());
    return 0;
}

// use_hook: take the renorm hook's host callback on this step.  Never while capturing: the callback enqueues a real collective
// (a rank whose graphs are already cached would not issue it: the all-reduces would pair up wrongly or hang) and the caller's
// `renorm_stats` pointer would be baked into the cached graph and outlive the hook.
static int enqueue_step(dsg_handle* h, const RunCtx& c, const UpdateArgs& u, bool renorm, hipStream_t s,
                        hipEvent_t* ev = nullptr, bool use_hook = true, int chunks = 1, size_t chunk_n = 0) {
    if (ev) {  // DSG_SAMPLE_PROFILE: one event pair per operator launch
        const bool fuse = h->fuse_hi - h->fuse_lo >= 2;
        bool skip_next = false;
        int tail = 0;
        for (size_t i = 0; i < h->ops.size(); ++i) {
            HIPCK(hipEventRecord(ev[2 * i], s));
            // the fused narrow run is one launch: its time is booked on its first operator, the others read ~0
            if (fuse && (int)i == h->fuse_lo) tail = launch_fused(h, c, s);
            else if (fuse && (int)i > h->fuse_lo && (int)i < h->fuse_hi + tail) {}
            else if (skip_next) { skip_next = false; }   // consumed by the pair launch booked on the previous operator
            else if (try_dual64(h, (int)i, c, s) || try_pair(h, (int)i, c, s)) skip_next = true;
            else launch_op(h, h->ops[i], c, s);
            HIPCK(hipEventRecord(ev[2 * i + 1], s));
        }
    } else {
        run_unet(h, c, s);
    }
    const unsigned ublocks = (unsigned)(((u.n + 3) / 4 + 255) / 256 < 2048 ? ((u.n + 3) / 4 + 255) / 256 : 2048);
    UpdateArgs ua = u;
    ua.record_y = renorm ? 0 : 1;
    hipLaunchKernelGGL(k_update, dim3(ublocks), dim3(256), 0, s, ua);
    if (renorm && h->renorm_fn && use_hook) {
        // sharded call that standardises with the statistics of the WHOLE batch (MSR.py:136-137 on the concatenation of all
        // ranks' rows): local moments -> the caller's all-reduce of 3 doubles (enqueued on this stream) -> apply
        hipLaunchKernelGGL(k_renorm_moments, dim3(kRedBlocks), dim3(256), 0, s, u.y, u.n, h->red, h->red + kRedBlocks);
        hipLaunchKernelGGL(k_renorm_moments_final, dim3(1), dim3(1), 0, s, h->red, h->red + kRedBlocks, u.n, h->renorm_stats);
        h->renorm_fn(h->renorm_user);
        hipLaunchKernelGGL(k_renorm_apply_stats, dim3(kRedBlocks), dim3(256), 0, s, u.y, u.n, h->renorm_stats);
        hipLaunchKernelGGL(k_record, dim3(ublocks), dim3(256), 0, s, u.y, u.n, u.cp, u.step_ptr);
    } else if (renorm) {   // the record is of the renormalised y (MSR.py:136-141): separate launches on these (at most 4) steps
        // one call: the handle's two partial rows; chunked: [chunks][kRedBlocks] twice, one grid row per chunk
        double* p1 = chunks > 1 ? h->red_chunks : h->red;
        double* p2 = chunks > 1 ? h->red_chunks + (size_t)chunks * kRedBlocks : h->red + kRedBlocks;
        const size_t cn = chunks > 1 ? chunk_n : (size_t)0;
        hipLaunchKernelGGL(k_renorm_sum, dim3(kRedBlocks, chunks), dim3(256), 0, s, (const float*)u.y, u.n, p1, p2, cn);
        hipLaunchKernelGGL(k_renorm_apply, dim3(kRedBlocks, chunks), dim3(256), 0, s, u.y, u.n, (const double*)p1, (const double*)p2, cn,
                           (const CallParams*)u.cp, (const int*)u.step_ptr);      // + the trajectory record of the renormalised y
    }
    HIPCK(hipGetLastError());
    if (ev) {
        HIPCK(hipStreamSynchronize(s));
        for (size_t i = 0; i < h->ops.size(); ++i) {
            float ms = 0.f;
            HIPCK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
            h->op_ms[i] += ms;
            h->op_calls[i] += 1;
        }
    }
    return 0;
}

int dsg_sample_rec(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed, float omega,
                   const float* coef, int T, float* out, int B, int flags, float* rec_y, float* rec_eps, void* stream);

int dsg_sample(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed, float omega,
               const float* coef, int T, float* out, int B, int flags, void* stream) {
    return dsg_sample_rec(h, cond, y_T, noise, seed, omega, coef, T, out, B, flags, nullptr, nullptr, stream);
}

namespace {
constexpr int kStepRunGraphs = 6;       // captured runs of 1, 2, 4, 8, 16, 32 later steps (dsg_handle::gexec)

// chunk_rows = 0: one call over B rows.  Otherwise the batch is ceil(B / chunk_rows) independent calls (own seed, own renorm
// statistics), chunk_rows a multiple of 32 so that every chunk starts on a row tile.
int sample_impl(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed, const unsigned long long* seeds_host,
                int chunk_rows, float omega, const float* coef, int T, float* out, int B, int flags, float* rec_y, float* rec_eps, hipStream_t s) {
    if (check_bound(h)) return 1;
    if (B < 1 || T < 1) return fail("B and T must be >= 1");
    if (!coef || !cond || !out) return fail("dsg_sample: null pointer argument");
    const int chunks = chunk_rows > 0 ? cdiv(B, chunk_rows) : 1;
    if (chunk_rows > 0 && (chunk_rows % 32 != 0)) return fail("dsg_sample_chunked: chunk_rows must be a multiple of 32 (a row tile)");
    if (chunks > 1 && h->renorm_fn) return fail("dsg_sample_chunked: a renorm hook (sharded global renorm) and chunking exclude each other");
    if (chunks > 1 && !seeds_host && !(y_T && (noise || T <= 2))) return fail("dsg_sample_chunked: per-chunk seeds are required");
    if (ensure_workspace(h, B, T)) return 1;
    const int D = h->d.input_dim, CG = groups_of(h->d.cond_dim), tpp = cdiv(B, 32);
    const size_t n = (size_t)B * D;
    const size_t chunk_n = chunks > 1 ? (size_t)chunk_rows * D : 0;
    if (chunks > 1) {
        if (chunks > h->seeds_cap) {
            HIPCK(hipStreamSynchronize(s));
            if (h->seeds_dev) (void)hipFree(h->seeds_dev);
            HIPCK(hipMalloc(&h->seeds_dev, (size_t)chunks * sizeof(unsigned long long)));
            h->seeds_cap = chunks;
        }
        if (chunks > h->red_chunks_cap) {
            HIPCK(hipStreamSynchronize(s));
            if (h->red_chunks) (void)hipFree(h->red_chunks);
            HIPCK(hipMalloc(&h->red_chunks, (size_t)2 * chunks * kRedBlocks * sizeof(double)));
            h->red_chunks_cap = chunks;
            free_graphs(h);                        // the captured renorm kernels hold the old buffer
        }
        std::vector<unsigned long long> sd(chunks, 0ull);
        if (seeds_host) sd.assign(seeds_host, seeds_host + chunks);
        HIPCK(hipMemcpyAsync(h->seeds_dev, sd.data(), (size_t)chunks * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        HIPCK(hipStreamSynchronize(s));            // `sd` is pageable host memory
    }

    // per-call setup: time table for all T steps, condition fragments, start state, step counter
    hipLaunchKernelGGL(k_linspace_t, dim3(cdiv(T, 256)), dim3(256), 0, s, h->tvals, T);
    run_time_path(h, T, s);
    hipLaunchKernelGGL(k_cond_frag, dim3(cdiv(tpp * CG * 256, 256)), dim3(256), 0, s, cond, (const float*)nullptr, B,
                       h->d.cond_dim, CG, h->condfrag, tpp);
    // condition embeddings of every block, once per call: cond is the same in all T steps (MSR.py:126-127)
    run_cond_embed(h, B, s);
    if (y_T) HIPCK(hipMemcpyAsync(h->ywork, y_T, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    else if (chunks > 1) hipLaunchKernelGGL(k_randn_chunked, dim3(2048), dim3(256), 0, s, h->ywork, n, (const unsigned long long*)h->seeds_dev, chunk_n / 4, 0xFFFFFFFFu);
    else hipLaunchKernelGGL(k_randn, dim3(2048), dim3(256), 0, s, h->ywork, n, seed, 0xFFFFFFFFu);
    // step counter (every step's first operator decrements it before anything reads it: step T-1 first) and the call block go
    // to the device as kernel arguments: no pageable copy, no synchronise - consecutive calls pipeline on the stream
    const CallParams cp{noise, coef, omega, T, seed, rec_y, rec_eps, chunk_n / 4, chunks > 1 ? h->seeds_dev : nullptr};
    hipLaunchKernelGGL(k_set_call, dim3(1), dim3(64), 0, s, h->step_dev, reinterpret_cast<CallParams*>(h->call_dev), cp, T);

    RunCtx c{B, 2, tpp, h->ywork, h->eps, h->step_dev, nullptr, false, true};
    c.advance_step = h->step_dev;
    c.share_proj = split_ctx(h, c);
    if (prepare_fused(h, c, s)) return 1;
    UpdateArgs u;
    u.eps = h->eps; u.y = h->ywork; u.cp = h->call_dev; u.step_ptr = h->step_dev; u.n = n; u.record_y = 0;

    const int n_renorm = T < 4 ? T : 4;  // steps i > T-5 (MSR.py:136)
    if (flags & DSG_SAMPLE_PROFILE) {
        const size_t nops = h->ops.size();
        h->op_ms.assign(nops, 0.0);
        h->op_calls.assign(nops, 0);
        std::vector<hipEvent_t> ev(2 * nops);
        for (auto& e : ev) HIPCK(hipEventCreate(&e));
        int rc = 0;
        for (int k = 0; k < T && !rc; ++k) rc = enqueue_step(h, c, u, k < n_renorm, s, ev.data(), true, chunks, chunk_n);
        for (auto& e : ev) (void)hipEventDestroy(e);
        if (rc) return 1;
    } else if (flags & DSG_SAMPLE_NO_GRAPH) {
        for (int k = 0; k < T; ++k)
            if (enqueue_step(h, c, u, k < n_renorm, s, nullptr, true, chunks, chunk_n)) return 1;
    } else {
        // The step index is a device counter and everything per call sits in device memory, so a captured step depends only on
        // the workspace (rows, chunk count) -- not on T, the seed or the schedule.  Captured once per workspace: one early
        // (renorm) step and runs of 1, 2, 4 ... 32 later steps; a call replays min(T, 4) early steps and the binary
        // decomposition of the rest (the shipped T = 20: 4 + one 16-step graph = 5 launches for the whole reverse loop).
        if (!(h->g_rows == B && h->g_chunks == chunks && h->g_chunk_rows == chunk_rows)) {
            free_graphs(h);
            h->g_rows = B; h->g_chunks = chunks; h->g_chunk_rows = chunk_rows;
        }
        // captured lazily: only the run lengths a call replays (a one-off batch size pays for its own decomposition, not for
        // 64 steps of nodes)
        auto graph_of = [&](int variant) -> int {
            if (h->gexec[variant]) return 0;
            const int run = variant == 0 ? 1 : 1 << (variant - 1);
            hipGraph_t g = nullptr;
            HIPCK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
            int rc = 0;
            for (int k = 0; k < run && !rc; ++k) rc = enqueue_step(h, c, u, variant == 0, h->cap_stream, nullptr, /*use_hook=*/false, chunks, chunk_n);
            hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
            if (rc) return 1;
            if (e != hipSuccess) return fail("hipStreamEndCapture: %s", hipGetErrorString(e));
            e = hipGraphInstantiate(&h->gexec[variant], g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) return fail("hipGraphInstantiate: %s", hipGetErrorString(e));
            return 0;
        };
        for (int k = 0; k < n_renorm; ++k) {
            if (h->renorm_fn) {                           // the hook calls back into the host: these (<= 4) steps run eagerly
                if (enqueue_step(h, c, u, true, s)) return 1;
            } else {
                if (graph_of(0)) return 1;
                HIPCK(hipGraphLaunch(h->gexec[0], s));
            }
        }
        for (int left = T - n_renorm; left > 0;) {
            int v = kStepRunGraphs - 1;
            while ((1 << v) > left) --v;
            if (graph_of(1 + v)) return 1;
            HIPCK(hipGraphLaunch(h->gexec[1 + v], s));
            left -= 1 << v;
        }
    }
    HIPCK(hipMemcpyAsync(out, h->ywork, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
}  // namespace

int dsg_sample_rec(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed, float omega,
                   const float* coef, int T, float* out, int B, int flags, float* rec_y, float* rec_eps, void* stream) {
    return sample_impl(h, cond, y_T, noise, seed, nullptr, 0, omega, coef, T, out, B, flags, rec_y, rec_eps, (hipStream_t)stream);
}

int dsg_sample_chunked(dsg_handle* h, const float* cond, const float* y_T, const float* noise, const unsigned long long* seeds, int chunk_rows,
                       float omega, const float* coef, int T, float* out, int B, int flags, void* stream) {
    if (chunk_rows < 1) return fail("dsg_sample_chunked: chunk_rows must be >= 1");
    return sample_impl(h, cond, y_T, noise, seeds ? seeds[0] : 0ull, seeds, chunk_rows, omega, coef, T, out, B, flags, nullptr, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------
// training step
// ------------------------------------------------------------------------------------------------------
namespace {
int train_step_impl(dsg_handle* h, const float* y, const float* cond, const int* ts, const float* noise, const float* cond_mask,
                    const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat, float* loss_out, int B, hipStream_t s);
}

int dsg_train_step(dsg_handle* h, const float* y, const float* cond, const int* ts, const float* noise, const float* cond_mask,
                   const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat, float* loss_out, int B, void* stream) {
    if (check_bound(h)) return 1;
    if (B < 1 || T < 1) return fail("B and T must be >= 1");
    if (!y || !cond || !ts || !noise || !sqrt_acp || !sqrt_1m_acp || !grads_flat || !loss_out)
        return fail("dsg_train_step: null pointer argument");
    if (ensure_train_workspace(h, B, T)) return 1;
    return train_step_impl(h, y, cond, ts, noise, cond_mask, sqrt_acp, sqrt_1m_acp, T, grads_flat, loss_out, B, (hipStream_t)stream);
}

int dsg_train_draws(unsigned long long seed, unsigned long long call, int T, float keep_prob, int B, int D, int* ts, float* noise,
                    float* cond_mask, void* stream) {
    if (B < 1 || D < 1 || T < 1) return fail("dsg_train_draws: B, D and T must be >= 1");
    if (call >> 30) return fail("dsg_train_draws: call counter out of range (2^30)");
    const size_t blocks = (((size_t)B * D + 3) / 4 + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    hipLaunchKernelGGL(k_train_draws, dim3(grid), dim3(256), 0, (hipStream_t)stream, ts, noise, cond_mask, B, D, T, keep_prob, seed, (unsigned)call,
                       (const unsigned long long*)nullptr);
    HIPCK(hipGetLastError());
    return 0;
}

int dsg_train_step_seeded_dyn(dsg_handle* h, const float* y, const float* cond, unsigned long long seed, unsigned long long* call_dev,
                              float keep_prob, const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat, float* loss_out,
                              int B, void* stream) {
    if (check_bound(h)) return 1;
    if (B < 1 || T < 1) return fail("B and T must be >= 1");
    if (!y || !cond || !call_dev || !sqrt_acp || !sqrt_1m_acp || !grads_flat || !loss_out) return fail("dsg_train_step_seeded_dyn: null pointer argument");
    if (ensure_train_workspace(h, B, T)) return 1;
    const int D = h->d.input_dim;
    if (!h->tr_noise) {
        HIPCK(hipMalloc(&h->tr_noise, (size_t)h->tr_rows * D * sizeof(float)));
        HIPCK(hipMalloc(&h->tr_mask, (size_t)h->tr_rows * sizeof(float)));
    }
    const size_t blocks = (((size_t)B * D + 3) / 4 + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 4096 ? blocks : 4096);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_train_draws, dim3(grid), dim3(256), 0, s, h->tr_ts, h->tr_noise, h->tr_mask, B, D, T, keep_prob, seed, 0u,
                       (const unsigned long long*)call_dev);
    hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(64), 0, s, call_dev);
    HIPCK(hipGetLastError());
    return train_step_impl(h, y, cond, h->tr_ts, h->tr_noise, h->tr_mask, sqrt_acp, sqrt_1m_acp, T, grads_flat, loss_out, B, s);
}

int dsg_train_step_seeded(dsg_handle* h, const float* y, const float* cond, unsigned long long seed, unsigned long long call,
                          float keep_prob, const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat, float* loss_out,
                          int B, void* stream) {
    if (check_bound(h)) return 1;
    if (B < 1 || T < 1) return fail("B and T must be >= 1");
    if (!y || !cond || !sqrt_acp || !sqrt_1m_acp || !grads_flat || !loss_out) return fail("dsg_train_step_seeded: null pointer argument");
    if (ensure_train_workspace(h, B, T)) return 1;
    const int D = h->d.input_dim;
    if (!h->tr_noise) {
        HIPCK(hipMalloc(&h->tr_noise, (size_t)h->tr_rows * D * sizeof(float)));
        HIPCK(hipMalloc(&h->tr_mask, (size_t)h->tr_rows * sizeof(float)));
    }
    // ts goes straight into the workspace copy the step reads (train_step_impl skips its own copy when handed that pointer)
    if (dsg_train_draws(seed, call, T, keep_prob, B, D, h->tr_ts, h->tr_noise, h->tr_mask, stream)) return 1;
    return train_step_impl(h, y, cond, h->tr_ts, h->tr_noise, h->tr_mask, sqrt_acp, sqrt_1m_acp, T, grads_flat, loss_out, B, (hipStream_t)stream);
}

namespace {
int train_step_impl(dsg_handle* h, const float* y, const float* cond, const int* ts, const float* noise, const float* cond_mask,
                    const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat, float* loss_out, int B, hipStream_t s) {
    // every shape the backward kernels cannot take is refused HERE, before anything is enqueued (the same check in the middle of
    // the enqueue left forked streams unjoined and the gradient bucket half written)
    if (h->use_split)
        for (const ResP& r : h->res)
            if (r.N >= 64 && (r.in0 != r.N || (r.in1 != 0 && r.in1 != r.N)))
                return fail("dsg_train_step: no backward kernel for a %d-wide block with inputs %d + %d wide", r.N, r.in0, r.in1);
    if (build_train_descs(h, B, T, s)) return 1;
    const int D = h->d.input_dim, C = h->d.cond_dim, CG = groups_of(C), DG = groups_of(D), tiles = cdiv(B, 32);
    const int td = h->td, half = h->d.proj_dim / 2;
    const float* A = h->arena;
    const Param* P = h->params.data();

    // ---- forward
    auto mark = [&](int i) { if (h->train_prof) (void)hipEventRecord(h->tev[i], s); };
    mark(0);
    if (ts != h->tr_ts) HIPCK(hipMemcpyAsync(h->tr_ts, ts, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
    // The time path forward (TimeEmbedding + the blocks' time_emb Linears over the T table rows: five short dependent launches) needs
    // nothing of the batch: once the side stream exists (large batches, from the second step on) it runs there, beside the draws'
    // consumers (condition fragments, q-sample, condition embeddings), and joins in front of the first block.
    const bool time_fwd_beside = h->side_stream && h->opt_time_beside && h->use_split;
    hipStream_t tstream = time_fwd_beside ? h->side_stream : s;
    if (time_fwd_beside) {
        HIPCK(hipEventRecord(h->ev_tail[0], s));                           // the weights of this step (optimizer step, re-pack) are in place
        HIPCK(hipStreamWaitEvent(h->side_stream, h->ev_tail[0], 0));
    }
    hipLaunchKernelGGL(k_linspace_t, dim3(cdiv(T, 256)), dim3(256), 0, tstream, h->tvals, T);
    run_time_path(h, T, tstream, true);
    if (time_fwd_beside) HIPCK(hipEventRecord(h->ev_tail[2], h->side_stream));
    hipLaunchKernelGGL(k_cond_frag, dim3(cdiv(tiles * CG * 256, 256)), dim3(256), 0, s, cond, cond_mask, B, C, CG, h->condfrag, tiles);
    hipLaunchKernelGGL(k_qsample, dim3(cdiv(tiles * DG * 256, 256)), dim3(256), 0, s, y, noise, h->tr_ts, sqrt_acp, sqrt_1m_acp, B, D,
                       h->tr_yt_rm, trp(h, h->tr_yt_frag), tiles);
    // the forward may run on the split-f16 kernels (they need the condition embeddings as an additive term)
    if (h->use_split) run_cond_embed(h, B, s);
    RunCtx c{B, 1, 0, h->tr_yt_rm, h->eps, nullptr, h->tr_ts, true, h->use_split};
    if (prepare_fused(h, c, s)) return 1;
    if (time_fwd_beside) HIPCK(hipStreamWaitEvent(s, h->ev_tail[2], 0));   // the time table
    run_unet(h, c, s);
    hipLaunchKernelGGL(k_loss_grad, dim3(kRedBlocks), dim3(256), 0, s, h->eps, noise, B, D, trp(h, h->tr_deps), tiles, h->red);
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(256), 0, s, h->red, kRedBlocks, (double)B * (double)D, loss_out);

    // ---- backward: activation gradients in reverse operator order
    mark(1);
    HIPCK(hipMemsetAsync(h->tr_gmax_t, 0, (size_t)h->n_gmax * h->gmax_ld * sizeof(unsigned), s));
    int next_part = 0;
    if (h->use_split && !h->wg_fork_ops.empty() && !h->side_stream) {
        HIPCK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
        HIPCK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_h), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        // (The step's tail runs on the side stream too, behind the early parts.  A separate high-priority stream for it measured 2 %
        // faster in a process of its own and 2.2x SLOWER -- 4.3 ms per step -- as soon as a second handle was alive in the process,
        // bench.py's sampling model: profiles/r04_train_tail_ab.txt.  No stream priorities.)
        for (auto& e : h->ev_tail) HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // the G and A operands of a part are final once the backward kernel of its last block is enqueued: its weight gradients run
    // on the side stream from there, beside the remaining chain kernels (one wave per SIMD below 65 536 rows: latency-bound,
    // half the CU idle).  Extra dynamic LDS keeps them to one workgroup per CU so that a chain kernel's workgroup always finds room.
    auto fork_parts = [&](int done_op) -> int {
        while (next_part < (int)h->wg_fork_ops.size() && h->wg_fork_ops[next_part] >= done_op) {
            const int u0 = next_part ? h->wg_part_end[next_part - 1] : 0, u1 = h->wg_part_end[next_part];
            unsigned* gm = h->tr_gmax + (size_t)(1 + next_part) * kMaxGmax;
            HIPCK(hipEventRecord(h->ev_fork, s));
            HIPCK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
            hipLaunchKernelGGL(k_gmax_reduce, dim3(h->n_gmax), dim3(256), 0, h->side_stream, h->tr_gmax_t, h->gmax_ld, gm);
            if (u1 > u0)
                hipLaunchKernelGGL(k_wgrad_h, dim3(u1 - u0), dim3(256), h->wg_early_lds, h->side_stream, h->wg_desc_dev, h->wg_unit_dev + u0,
                                   gm, h->tr_slabs, h->slab_stride, tiles, h->tr_chunks);
            ++next_part;
        }
        return 0;
    };
    for (int oi = (int)h->ops.size() - 1; oi >= 0; --oi) {
        const Op& op = h->ops[oi];
        if (op.kind == OP_PROJ) continue;
        if (h->use_split && h->fbwd_n > 0 && oi == h->fuse_hi - 1) {
            // the narrow run: one launch walks its operators in reverse (k_fused_narrow_bwd_h)
            hipLaunchKernelGGL(k_fused_narrow_bwd_h, dim3(cdiv(tiles, kWavesPerBlock)), dim3(256), 0, s, h->fbwd_dev, h->fbwd_n, tiles);
            oi = h->fuse_lo;
            if (fork_parts(oi)) return 1;
            continue;
        }
        if (op.kind == OP_RES) {
            const ResP& r = h->res[op.p];
            BlockBwdArgs a;
            BlockBwdArgsH ah;
            fill_res_bwd_args(h, op, tiles, a, ah);
            if (h->use_split) {
                if (!launch_res_bwd_h(r.N, r.sclin, ah, s, true))
                    return fail("dsg_train_step: no backward kernel for a %d-wide block with inputs %d + %d wide", r.N, r.in0, r.in1);
                if (fork_parts(oi)) return 1;
            } else {
                launch_res_bwd(r.N, r.sclin, a, s);
            }
        } else {
            const LinOpP& l = h->lin[op.p];
            LinBwdArgs a;
            fill_lin_bwd_args(h, op, tiles, a);
            launch_lin_bwd(l.l.K, op.kind == OP_FINAL, a, s);
        }
    }
    // ---- weight / bias / LayerNorm gradients: grouped launches into per-chunk slabs, then a fixed-order reduce.
    // Small batches: everything in order on the caller's stream.  With the side stream (>= 32 768 rows) the tail runs on two streams:
    //   caller's stream       max|G| of the blocks -> the blocks' units (ONE launch, longest first) -> the Linears' units -> reduce (parameters)
    //   side stream (behind   time-table units -> column sums (k_cs_reduce, k_colsum) -> max|G| of everything --^                ^
    //   the early parts)      -> reduce (dTB) -> time path ----------------------------------------------------------------------+
    // A residual block's G operands have their scale from the block's own backward kernel; only the plain Linears' wait for k_colsum.
    // The time path (UNetCF.py:35-44 backward: ~10 short dependent launches) needs nothing but the dTB rows; its five results go
    // straight into the caller's gradient bucket, and the fixed-order reduce at the end covers every OTHER parameter
    // (k_reduce_ranges: the time_emb weights are ~40 % of the parameters and nothing in the slabs belongs to them).
    mark(2);
    const bool time_beside = h->use_split && next_part > 0 && h->side_stream && h->opt_time_beside;
    hipStream_t cs_stream = time_beside ? h->side_stream : s;
    unsigned* const gm_all = time_beside ? h->tr_gmax + (size_t)(1 + kMaxWgParts) * kMaxGmax : h->tr_gmax;
    auto units = [&](int lo, int hi, const unsigned* gm, hipStream_t us) {
        if (hi > lo)
            hipLaunchKernelGGL(k_wgrad_h, dim3(hi - lo), dim3(256), 0, us, h->wg_desc_dev, h->wg_unit_dev + lo, gm, h->tr_slabs, h->slab_stride,
                               tiles, h->tr_chunks);
    };
    const int u0 = next_part ? h->wg_part_end[next_part - 1] : 0;
    if (time_beside) {
        // the blocks' slots are complete; the Linears' are not yet, and nothing reads them from this set
        hipLaunchKernelGGL(k_gmax_reduce, dim3(h->n_gmax), dim3(256), 0, s, h->tr_gmax_t, h->gmax_ld, h->tr_gmax);
        HIPCK(hipEventRecord(h->ev_tail[0], s));
        HIPCK(hipStreamWaitEvent(h->side_stream, h->ev_tail[0], 0));
        units(u0, h->wg_onehot_end, h->tr_gmax, h->side_stream);
    }
    hipLaunchKernelGGL(k_cs_reduce, dim3(cdiv(h->cs_slots, 256), h->tr_chunks), dim3(256), 0, cs_stream, h->tr_cs, h->cs_map_dev, h->cs_slots,
                       h->tr_slabs, h->slab_stride, tiles, h->tr_chunks);
    hipLaunchKernelGGL(k_colsum, dim3(cdiv(h->cs_units, 4)), dim3(256), 0, cs_stream, h->cs_desc_dev, h->cs_unit_dev, h->cs_units, h->tr_slabs,
                       h->slab_stride, tiles, h->tr_chunks, B, h->tr_gmax_t, h->gmax_ld);
    hipLaunchKernelGGL(k_gmax_reduce, dim3(h->n_gmax), dim3(256), 0, cs_stream, h->tr_gmax_t, h->gmax_ld, gm_all);
    mark(3);
    const size_t dtb0 = (size_t)h->total_params, dtb_n = h->slab_stride - dtb0;
    auto reduce_blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)); };
    auto time_path = [&](float* G, hipStream_t ts) {
        float* emb = h->tr_tsave;
        float* h1pre = emb + (size_t)T * 2 * half;
        float* h1s = h1pre + (size_t)T * td;
        float* tpre = h1s + (size_t)T * td;
        float* d_st = tpre + (size_t)T * td;
        float* d_h1s = d_st + (size_t)T * td;
        float* tpart = d_h1s + (size_t)T * td;   // [kTimeChunks][T][td]
        // all blocks at once: their dTB rows are contiguous from the first block's dtb_off
        const float* dtb_all = h->tr_gsum + h->res[0].dtb_off;
        hipLaunchKernelGGL(k_time_wgrad, dim3(2048), dim3(256), 0, ts, dtb_all, T, h->st, td, h->tw_dst_dev, G, h->tw_rows);
        hipLaunchKernelGGL(k_time_dgrad, dim3(cdiv(T * td, 256), kTimeChunks), dim3(256), 0, ts, dtb_all, T, h->tw_src_dev, td, tpart, h->tw_rows);
        const unsigned eb = (unsigned)cdiv(T * td, 256);
        hipLaunchKernelGGL(k_time_dgrad_finish, dim3(eb), dim3(256), 0, ts, tpart, tpre, d_st, (size_t)T * td);  // d temb (pre-Swish)
        small_gemm(d_st, 1, td, h1s, td, 1, G + P[h->temb_l2w].off, td, 1, td, td, T, 0, ts);                   // d lin2.weight
        hipLaunchKernelGGL(k_col_sum_small, dim3(cdiv(td, 256)), dim3(256), 0, ts, d_st, T, td, (long long)td, G + P[h->temb_l2b].off);
        small_gemm(d_st, td, 1, P[h->temb_l2w].ptr, td, 1, d_h1s, td, 1, T, td, td, 0, ts);                     // d h1 (post-Swish)
        hipLaunchKernelGGL(k_mul_silu_grad, dim3(eb), dim3(256), 0, ts, d_h1s, h1pre, (size_t)T * td);
        small_gemm(d_h1s, 1, td, emb, 2 * half, 1, G + P[h->temb_l1w].off, 2 * half, 1, td, 2 * half, T, 0, ts);  // d lin1.weight
        hipLaunchKernelGGL(k_col_sum_small, dim3(cdiv(td, 256)), dim3(256), 0, ts, d_h1s, T, td, (long long)td, G + P[h->temb_l1b].off);
    };
    if (time_beside) {
        units(h->wg_onehot_end, h->wg_blocks_end, h->tr_gmax, s);
        HIPCK(hipEventRecord(h->ev_tail[2], h->side_stream));              // column sums done: the Linears' scales
        HIPCK(hipStreamWaitEvent(s, h->ev_tail[2], 0));
        units(h->wg_blocks_end, h->wg_units, gm_all, s);
        hipLaunchKernelGGL(k_reduce_slabs, reduce_blocks(dtb_n), dim3(256), 0, h->side_stream, h->tr_slabs + dtb0, h->slab_stride, h->tr_chunks,
                           h->tr_gsum + dtb0, dtb_n);
        time_path(grads_flat, h->side_stream);       // straight into the caller's bucket: the last reduce leaves these regions out
        HIPCK(hipEventRecord(h->ev_tail[1], h->side_stream));
        HIPCK(hipStreamWaitEvent(s, h->ev_tail[1], 0));
        mark(4);
        if (h->r2_vec4 && (reinterpret_cast<unsigned long long>(grads_flat) & 15ull) == 0)
            hipLaunchKernelGGL(k_reduce_ranges4, reduce_blocks((size_t)h->r2_total / 4), dim3(256), 0, s, h->tr_slabs, h->slab_stride, h->tr_chunks,
                               grads_flat, (const long long*)(h->r2_dev + 2 * h->r2_n), (const long long*)(h->r2_dev + 3 * h->r2_n), h->r2_n,
                               h->r2_total / 4);
        else
            hipLaunchKernelGGL(k_reduce_ranges, reduce_blocks((size_t)h->r2_total), dim3(256), 0, s, h->tr_slabs, h->slab_stride, h->tr_chunks,
                               grads_flat, (const long long*)h->r2_dev, (const long long*)(h->r2_dev + h->r2_n), h->r2_n, h->r2_total);
    } else {
        if (h->use_split) {
            if (next_part) HIPCK(hipEventRecord(h->ev_join, h->side_stream));
            units(u0, h->wg_units, h->tr_gmax, s);
            if (next_part) HIPCK(hipStreamWaitEvent(s, h->ev_join, 0));
        } else
            hipLaunchKernelGGL(k_wgrad, dim3(h->wg_units), dim3(256), 0, s, h->wg_desc_dev, h->wg_unit_dev, h->tr_slabs, h->slab_stride, tiles,
                               h->tr_chunks);
        mark(4);
        hipLaunchKernelGGL(k_reduce_slabs, reduce_blocks(h->slab_stride), dim3(256), 0, s, h->tr_slabs, h->slab_stride, h->tr_chunks,
                           h->tr_gsum, h->slab_stride);
        time_path(h->tr_gsum, s);
        HIPCK(hipMemcpyAsync(grads_flat, h->tr_gsum, (size_t)h->total_params * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    mark(5);
    if (h->train_prof) h->tev_valid = true;
    HIPCK(hipGetLastError());
    return 0;
}
}  // namespace

int dsg_train_profile_enable(dsg_handle* h, int on) {
    if (!h) return fail("null handle");
    if (on && !h->tev[0])
        for (auto& e : h->tev) HIPCK(hipEventCreate(&e));
    h->train_prof = on != 0;
    h->tev_valid = false;
    return 0;
}

int dsg_train_profile(dsg_handle* h, float* ms5) {
    if (!h || !ms5) return fail("dsg_train_profile: null argument");
    if (!h->tev_valid) return fail("dsg_train_profile: no profiled dsg_train_step yet (dsg_train_profile_enable first)");
    HIPCK(hipEventSynchronize(h->tev[5]));
    for (int i = 0; i < 5; ++i) HIPCK(hipEventElapsedTime(&ms5[i], h->tev[i], h->tev[i + 1]));
    return 0;
}

int dsg_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int maximize, long long step, void* stream) {
    if (!p || !g || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return fail("dsg_adam_step: bad arguments");
    if (n == 0) return 0;
    if ((reinterpret_cast<unsigned long long>(p) | reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(exp_avg) |
         reinterpret_cast<unsigned long long>(exp_avg_sq)) & 15ull)
        return fail("dsg_adam_step: the four buffers must be 16-byte aligned");
    const AdamArgs a{p, g, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, weight_decay, eps, (float)step, maximize, nullptr, nullptr};
    const long long pieces = (n / 4 + 255) / 256;
    const unsigned blocks = (unsigned)(pieces < 1 ? 1 : (pieces < 2048 ? pieces : 2048));
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    HIPCK(hipGetLastError());
    return 0;
}

int dsg_adam_step_dyn(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, const double* lr_dev, double beta1, double beta2,
                      double eps, double weight_decay, int maximize, float* step_dev, void* stream) {
    if (!p || !g || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev || n < 0) return fail("dsg_adam_step_dyn: bad arguments");
    if (n == 0) return 0;
    if ((reinterpret_cast<unsigned long long>(p) | reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(exp_avg) |
         reinterpret_cast<unsigned long long>(exp_avg_sq)) & 15ull)
        return fail("dsg_adam_step_dyn: the four buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_bump_f32, dim3(1), dim3(64), 0, s, step_dev);          // the count AFTER this update, as dsg_adam_step is handed it
    const AdamArgs a{p, g, exp_avg, exp_avg_sq, (size_t)n, 0.0, beta1, beta2, weight_decay, eps, 0.f, maximize, lr_dev, step_dev};
    const long long pieces = (n / 4 + 255) / 256;
    const unsigned blocks = (unsigned)(pieces < 1 ? 1 : (pieces < 2048 ? pieces : 2048));
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, a);
    HIPCK(hipGetLastError());
    return 0;
}

int dsg_ema_update(float* avg, const float* p, float decay, float one_minus_decay, long long n, void* stream) {
    if (!avg || !p || n < 0) return fail("dsg_ema_update: bad arguments");
    if (n == 0) return 0;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_ema, dim3(blocks), dim3(256), 0, (hipStream_t)stream, avg, p, decay, one_minus_decay, (size_t)n);
    HIPCK(hipGetLastError());
    return 0;
}

}  // extern "C"

// ---- decoders / evaluators (dsg_eval.hpp)
namespace {
int eval_args(const void* a, const void* b, long long rows, int D, const char* who) {
    if (!a || !b || rows < 0 || D < 1) return fail("%s: bad arguments", who);
    return 0;
}
// part[0] <- (min, max) of columns [c0, c1) of the whole tensor; stream-ordered scratch, freed by the caller after its consumer
int minmax_global(const float* y, long long rows, int D, int c0, int c1, float2** part, hipStream_t s) {
    const int w = c1 - c0, lpr = w >= 64 ? 64 : (w >= 16 ? 16 : (w >= 4 ? 4 : 1));
    const long long want = (rows + 4LL * (64 / lpr) - 1) / (4LL * (64 / lpr));   // one trip of k_minmax_partial per block if it fits
    const int nparts = (int)(want < kEvalParts ? want : kEvalParts);
    HIPCK(hipMallocAsync(reinterpret_cast<void**>(part), (size_t)(1 + nparts) * sizeof(float2), s));
    hipLaunchKernelGGL(k_minmax_partial, dim3(nparts), dim3(256), 0, s, y, rows, D, c0, c1, *part);
    hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, s, *part, nparts);
    return 0;
}
template <int MODE>
int softmax_rows(const float* y, float* out, long long rows, int D, hipStream_t s, int c0, int c1) {
    if (rows == 0) return 0;
    float2* part = nullptr;
    if (MODE == 1 && minmax_global(y, rows, D, c0, c1, &part, s)) return 1;
    auto blocks = [&](int lpr) { const long long rpb = 4LL * (64 / lpr); return dim3((unsigned)((rows + rpb - 1) / rpb)); };
    if (D <= kSoftEpl) hipLaunchKernelGGL((k_row_softmax<MODE, 1>), blocks(1), dim3(256), 0, s, y, out, rows, D, part);
    else if (D <= 4 * kSoftEpl) hipLaunchKernelGGL((k_row_softmax<MODE, 4>), blocks(4), dim3(256), 0, s, y, out, rows, D, part);
    else if (D <= 16 * kSoftEpl) hipLaunchKernelGGL((k_row_softmax<MODE, 16>), blocks(16), dim3(256), 0, s, y, out, rows, D, part);
    else if (D <= 64 * kSoftEpl) hipLaunchKernelGGL((k_row_softmax<MODE, 64>), blocks(64), dim3(256), 0, s, y, out, rows, D, part);
    else hipLaunchKernelGGL((k_row_softmax_wide<MODE>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, y, out, rows, D, part);
    if (part) HIPCK(hipFreeAsync(part, s));
    HIPCK(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" {

int dsg_row_softmax(const float* y, float* out, long long rows, int D, void* stream) {
    if (eval_args(y, out, rows, D, "dsg_row_softmax")) return 1;
    return softmax_rows<0>(y, out, rows, D, (hipStream_t)stream, 0, D);
}
int dsg_msr_decode(const float* y, float* out, long long rows, int D, void* stream) {
    if (eval_args(y, out, rows, D, "dsg_msr_decode")) return 1;
    return softmax_rows<1>(y, out, rows, D, (hipStream_t)stream, 0, D);
}
int dsg_co_decode(const float* y, float* out, long long rows, int D, void* stream) {
    if (eval_args(y, out, rows, D, "dsg_co_decode")) return 1;
    return softmax_rows<2>(y, out, rows, D, (hipStream_t)stream, 0, D);
}
int dsg_msr_rate(const float* p, const float* gain, float* rate, long long rows, int D, void* stream) {
    if (eval_args(p, gain, rows, D, "dsg_msr_rate") || !rate) return fail("dsg_msr_rate: bad arguments");
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (D <= 8) hipLaunchKernelGGL((k_msr_rate<1>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, p, gain, rate, rows, D);
    else if (D <= 160) hipLaunchKernelGGL((k_msr_rate<16>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, p, gain, rate, rows, D);
    else hipLaunchKernelGGL((k_msr_rate<64>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p, gain, rate, rows, D);
    HIPCK(hipGetLastError());
    return 0;
}
int dsg_co_cost(const float* X, const float* Y, float* cost, long long rows, int n, void* stream) {
    if (eval_args(X, Y, rows, n, "dsg_co_cost") || !cost) return fail("dsg_co_cost: bad arguments");
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_co_cost, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, Y, cost, rows, n);
    HIPCK(hipGetLastError());
    return 0;
}
int dsg_nu_decode(const float* y, float* out, long long rows, int D, float width, float height, float p_sum, void* stream) {
    if (eval_args(y, out, rows, D, "dsg_nu_decode")) return 1;
    if (D < 3) return fail("dsg_nu_decode: D = %d (two position columns and at least one power column)", D);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    float2* part = nullptr;
    if (minmax_global(y, rows, D, 0, 2, &part, s)) return 1;
    hipLaunchKernelGGL(k_nu_decode, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, y, out, rows, D, width, height, p_sum, part);
    HIPCK(hipFreeAsync(part, s));
    HIPCK(hipGetLastError());
    return 0;
}
int dsg_nu_rate(const float* Yd, const float* X, float* rate, long long rows, int K, void* stream) {
    if (eval_args(Yd, X, rows, K, "dsg_nu_rate") || !rate) return fail("dsg_nu_rate: bad arguments");
    if (K > kNuMaxUsers) return fail("dsg_nu_rate: K = %d users (at most %d)", K, kNuMaxUsers);
    if (rows == 0) return 0;
    hipLaunchKernelGGL(k_nu_rate, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Yd, X, rate, rows, K);
    HIPCK(hipGetLastError());
    return 0;
}

int dsg_sum_rate_gen(const double* gs, double* schemes, double* rates, long long rows, int M, double W, void* stream) {
    if (!gs || !schemes || !rates || rows < 0 || M < 1) return fail("dsg_sum_rate_gen: bad arguments");
    if (M > kSrMaxM) return fail("dsg_sum_rate_gen: M = %d channels (at most %d)", M, kSrMaxM);
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int kMaxIter = 149;                       // the reference leaves its loop when k reaches 150
    int* flags = nullptr;
    HIPCK(hipMallocAsync(reinterpret_cast<void**>(&flags), (kMaxIter + 3) * sizeof(int), s));
    HIPCK(hipMemsetAsync(flags, 0, (kMaxIter + 3) * sizeof(int), s));
    HIPCK(hipMemsetAsync(flags, 1, 1, s));          // flags[0] = 1 (low byte): the dry pass always runs
    const long long n = rows * M;
    hipLaunchKernelGGL(k_sumrate_init, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, s, schemes, n, W / M);
    const dim3 grid((unsigned)((rows + 3) / 4));
    const double eps = 0.001;
    double beta = 0.1;
    hipLaunchKernelGGL(k_sumrate_iter, grid, dim3(256), 0, s, gs, schemes, rows, M, 0.0, eps, flags, flags + 1, 1);
    int k = 1;
    for (int it = 1; it <= kMaxIter; ++it) {
        hipLaunchKernelGGL(k_sumrate_iter, grid, dim3(256), 0, s, gs, schemes, rows, M, beta, eps, flags + it, flags + it + 1, 0);
        ++k;
        if (k % 20 == 0) beta *= 0.5;
        if (k == 150) break;
    }
    hipLaunchKernelGGL(k_sumrate_rates, grid, dim3(256), 0, s, gs, schemes, rates, rows, M);
    HIPCK(hipFreeAsync(flags, s));
    HIPCK(hipGetLastError());
    return 0;
}

int dsg_co_minlp_search(const double* params, const double* choices, int nch, double* Y, int* tolerable, long long rows, int n,
                        double F_t, double P_t, double P_I, double theta, void* stream) {
    if (!params || !choices || !Y || !tolerable) return fail("dsg_co_minlp_search: null pointer argument");
    if (n < 1 || n > kCoMaxNodes) return fail("dsg_co_minlp_search: node_num must be in [1, %d] (got %d)", kCoMaxNodes, n);
    if (nch < 1 || nch > 4096) return fail("dsg_co_minlp_search: bad grid size %d", nch);
    if (rows < 0 || rows > 0x7fffffffLL) return fail("dsg_co_minlp_search: bad row count");
    if (rows == 0) return 0;
    const CoGenConst cc{F_t, P_t, P_I, theta};
    hipLaunchKernelGGL(k_co_minlp, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, params, choices, nch, n, cc, Y, tolerable);
    HIPCK(hipGetLastError());
    return 0;
}

int dsg_op_count(const dsg_handle* h) { return h ? (int)h->ops.size() : 0; }

int dsg_fused_range(const dsg_handle* h, int* lo, int* hi) {
    if (!h) return fail("null handle");
    const bool on = h->fuse_hi - h->fuse_lo >= 2;
    if (lo) *lo = on ? h->fuse_lo : 0;
    if (hi) *hi = on ? h->fuse_hi : 0;
    return 0;
}

int dsg_op_profile(const dsg_handle* h, int op, double* ms_total, int* calls) {
    if (!h || op < 0 || op >= (int)h->op_ms.size()) return fail("dsg_op_profile: no profile for op %d (run dsg_sample with DSG_SAMPLE_PROFILE)", op);
    if (ms_total) *ms_total = h->op_ms[op];
    if (calls) *calls = h->op_calls[op];
    return 0;
}

int dsg_op_info(const dsg_handle* h, int op, char* name, double* flops_per_row, double* bytes_per_row) {
    if (!h || op < 0 || op >= (int)h->ops.size()) return fail("bad op index");
    const Op& o = h->ops[op];
    double macs2 = 0, macs_cond = 0, bytes = 0;  // macs2: both passes; macs_cond: conditional pass only
    if (o.kind == OP_RES) {
        const ResP& r = h->res[o.p];
        const int in = r.in0 + r.in1;
        macs2 = (double)in * r.N + 2.0 * r.N * r.N + (r.sclin ? (double)in * r.N : 0.0);
        macs_cond = (double)h->d.cond_dim * r.N;
        bytes = 2.0 * 4.0 * (in + r.N);
    } else {
        const LinOpP& l = h->lin[o.p];
        macs2 = (double)l.l.K * l.l.N;
        bytes = 2.0 * 4.0 * (l.l.K + l.l.N);
    }
    if (name) { strncpy(name, o.name.c_str(), 63); name[63] = 0; }
    (void)macs_cond;  // hoisted out of the step: the condition embeddings are computed once per dsg_sample call
    if (flops_per_row) *flops_per_row = 2.0 * (2.0 * macs2);
    if (bytes_per_row) *bytes_per_row = bytes;
    return 0;
}

int dsg_time_op(dsg_handle* h, int op, int B, int iters, float* ms_avg, void* stream) {
    if (check_bound(h)) return 1;
    if (op < 0 || op >= (int)h->ops.size() || iters < 1 || !ms_avg) return fail("dsg_time_op: bad arguments");
    if (ensure_workspace(h, B, 1)) return 1;
    hipStream_t s = (hipStream_t)stream;
    const int zero = 0;
    HIPCK(hipMemcpyAsync(h->step_dev, &zero, sizeof(int), hipMemcpyHostToDevice, s));
    RunCtx c{B, 2, cdiv(B, 32), h->ywork, h->eps, h->step_dev, nullptr, false, true};
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0));
    HIPCK(hipEventCreate(&e1));
    launch_op(h, h->ops[op], c, s);  // warm-up
    HIPCK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_op(h, h->ops[op], c, s);
    HIPCK(hipEventRecord(e1, s));
    HIPCK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_avg = ms / iters;
    return 0;
}

// ---- box calibration (bench.py `box`): the pool's MI355X boxes differ by up to 15 % on the same binary (clock under load), so a
// headline alone cannot tell a slow box from a slow tree.  Three fixed probes, timed with HIP events on `stream`, median of 5 launches
// of ~5 ms each (a 1-ms probe read 1 816 and 1 564 TFLOP/s on one box a second apart: the clock had not settled):
//   out[0] mfma_tflops  every SIMD issues dependent v_mfma_f32_32x32x16_f16 back to back on non-trivial operands (two waves per SIMD):
//                       the bare matrix pipe at the clock the box holds under it;
//   out[1] mix_gslots   the same loop with six vector instructions (one of them transcendental) behind every MFMA -- the instruction mix
//                       of the block kernels -- in 1e9 (MFMA + 6 vector) slots per second over the chip;
//   out[2] copy_gbs     a 256 MiB float4 copy inside a buffer allocated for the call (read + written bytes per second);
//   out[3] panel_gslots the frozen miniature of the panel kernels' load profile (k_calib_panel), 1e9 MFMA slots per second.
int dsg_box_calibrate(float* out3 /* four floats */, void* stream) {
    if (!out3) return fail("dsg_box_calibrate: null output");
    hipStream_t s = (hipStream_t)stream;
    hipDeviceProp_t prop;
    int devid = 0;
    HIPCK(hipGetDevice(&devid));
    HIPCK(hipGetDeviceProperties(&prop, devid));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float* sink = nullptr;
    const size_t copy_bytes = (size_t)256 << 20;
    char* buf = nullptr;
    struct Guard {          // every early return below (a failed 512 MiB allocation beside a large batch) frees what exists
        hipEvent_t &a, &b; float*& s; char*& c;
        ~Guard() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); if (s) (void)hipFree(s); if (c) (void)hipFree(c); }
    } guard{e0, e1, sink, buf};
    HIPCK(hipEventCreate(&e0));
    HIPCK(hipEventCreate(&e1));
    HIPCK(hipMalloc(&sink, (size_t)cus * 2 * 512 * sizeof(float)));
    HIPCK(hipMalloc(&buf, 2 * copy_bytes));
    hipLaunchKernelGGL(dsg::k_calib_fill, dim3(cus * 8), dim3(256), 0, s, reinterpret_cast<uint4*>(buf), 2 * copy_bytes / 16);
    auto median5 = [&](auto&& launch, float& ms_out) -> int {
        float t[6];
        for (int rep = 0; rep < 6; ++rep) {      // the first launch is a warm-up
            HIPCK(hipEventRecord(e0, s));
            launch();
            HIPCK(hipEventRecord(e1, s));
            HIPCK(hipEventSynchronize(e1));
            HIPCK(hipEventElapsedTime(&t[rep], e0, e1));
        }
        std::sort(t + 1, t + 6);
        ms_out = t[3];
        return 0;
    };
    // two workgroups of four waves per CU = two waves per SIMD, as the block kernels run
    const int it0 = 3072, it6 = 1536;             // x 48 MFMAs: ~5 ms each at two waves per SIMD
    float ms0 = 0.f, ms6 = 0.f, msc = 0.f;
    if (median5([&] { hipLaunchKernelGGL(dsg::k_calib_mfma<0>, dim3(2 * cus), dim3(256), 0, s, it0, sink); }, ms0)) return 1;
    if (median5([&] { hipLaunchKernelGGL(dsg::k_calib_mfma<6>, dim3(2 * cus), dim3(256), 0, s, it6, sink); }, ms6)) return 1;
    if (median5([&] { hipLaunchKernelGGL(dsg::k_calib_copy, dim3(cus * 8), dim3(256), 0, s, reinterpret_cast<const uint4*>(buf),
                                         reinterpret_cast<uint4*>(buf + copy_bytes), copy_bytes / 16); }, msc)) return 1;
    out3[0] = (float)((double)cus * 8 * it0 * 48 * 32768.0 / (ms0 * 1e-3) / 1e12);
    out3[1] = (float)((double)cus * 8 * it6 * 48 / (ms6 * 1e-3) / 1e9);
    out3[2] = (float)(2.0 * copy_bytes / (msc * 1e-3) / 1e9);
    // out3[3]: the frozen miniature of the panel kernels' load profile (dsg_panel.hpp, k_calib_panel): one workgroup per CU, ~10 ms per
    // launch -- long enough for the clock to settle under it -- 1e9 MFMA slots per second over the chip
    {
        const int panels = 1536;
        float msp = 0.f;
        if (median5([&] { hipLaunchKernelGGL(dsg::k_calib_panel, dim3(cus), dim3(512), 0, s, reinterpret_cast<const uint4*>(buf),
                                             reinterpret_cast<const uint4*>(buf + copy_bytes), copy_bytes / 16 - 1, panels, sink); }, msp)) return 1;
        out3[3] = (float)((double)cus * 8 * panels * 48 / (msp * 1e-3) / 1e9);
    }
    return 0;
}

}  // extern "C"

#ifdef DSG_CYCLE_STAMPS
// measurement builds only (tools/cycle_stamps.sh): the stamps recorded since the last call, oldest first by slot
extern "C" int dsg_stamps_fetch(unsigned long long* out, int n) {
    int cnt = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&cnt, HIP_SYMBOL(dsg::dsg_stamp_n), sizeof(int)) != hipSuccess) return -1;
    if (cnt > 8192) cnt = 8192;
    if (cnt > n) cnt = n;
    if (cnt > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(dsg::dsg_stamp_buf), (size_t)cnt * sizeof(unsigned long long)) != hipSuccess) return -1;
    const int zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dsg::dsg_stamp_n), &zero, sizeof(int));
    return cnt;
}
#endif

