// Host side of libdiffsg_hip.so: builds the operator plan of UNet1D (UNetCF.py:262-356), owns the packed-weight
// arena and the fragment-layout workspace, and enqueues the kernels of dsg_kernels.hpp.  C ABI: include/diffsg.h.
#include "dsg_kernels.hpp"
#include "../../include/diffsg.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>
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
    // packed (arena offsets, floats)
    size_t W1p, g1p, b1p, W2p, g2p, b2p, c2p, Wcp, W3p, g3p, b3p, c3p, Wscp;
};

struct LinOpP {            // feature_proj / Down/Upsample / final
    LinearP l;
    NormP ln;              // final only
    bool lnact = false;
    size_t Wp, bp, gp, betap;
};

struct TensorInfo { int width; size_t data_off, stats_off; };  // per-tile float offsets

enum OpKind { OP_PROJ, OP_RES, OP_LIN, OP_FINAL };
struct Op {
    OpKind kind;
    int p;          // index into res / lin
    int in0, in1;   // tensor ids (-1: none)
    int out;        // tensor id (-1 for final)
    std::string name;
};

}  // namespace

struct dsg_handle {
    dsg_unet_desc d;
    int td = 0;  // time_dim = 4*proj
    std::vector<Param> params;
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
    bool bound = false;

    // workspace
    int cap_rows = 0, cap_entries = 0;
    float* ws = nullptr;        // activations
    float* condfrag = nullptr;
    float* tb = nullptr;        // [entries][tb_stride]
    float* st = nullptr;        // [entries][td]
    float* tvals = nullptr;     // [entries]
    int* ts_ident = nullptr;    // [rows] identity index
    float* eps = nullptr;       // [2][rows][D]
    float* ywork = nullptr;     // [rows][D]
    float* freq = nullptr;      // [proj/2]
    double* red = nullptr;      // [2][kRedBlocks]
    int* step_dev = nullptr;
    CallParams* call_dev = nullptr;
    std::vector<double> op_ms;   // DSG_SAMPLE_PROFILE: summed HIP-event time per op
    std::vector<int> op_calls;
    hipStream_t cap_stream = nullptr;  // capture-only stream (the caller's may be the null stream)

    // cached step graphs
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    int g_rows = -1;
};

namespace {

int add_param(dsg_handle* h, const std::string& name, long long numel) {
    h->params.push_back(Param{name, numel});
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
int add_tensor(dsg_handle* h, int width) {
    TensorInfo t;
    t.width = width;
    t.data_off = h->per_tile_floats;
    h->per_tile_floats += (size_t)groups_of(width) * 256;
    t.stats_off = h->per_tile_floats;
    h->per_tile_floats += 64;
    h->tensors.push_back(t);
    return (int)h->tensors.size() - 1;
}

bool width_supported(int n) { return n == 4 || n == 8 || n == 16 || n == 32 || n == 64 || n == 128; }

// Arena carving -----------------------------------------------------------------------------------------
struct Carver {
    size_t off = 0;
    size_t take(size_t floats) { size_t o = off; off += (floats + 63) / 64 * 64; return o; }
};

void carve(dsg_handle* h) {
    Carver c;
    const int CG = groups_of(h->d.cond_dim);
    for (auto& r : h->res) {
        const int NT = cdiv(r.N, 32), NG = groups_of(r.N), KG = groups_of(r.in0) + groups_of(r.in1);
        r.W1p = c.take((size_t)NT * KG * 256);
        r.g1p = c.take((size_t)KG * 8 + 32);
        r.b1p = c.take((size_t)KG * 8 + 32);
        r.W2p = c.take((size_t)NT * NG * 256);
        r.g2p = c.take(NT * 32); r.b2p = c.take(NT * 32); r.c2p = c.take(NT * 32);
        r.Wcp = c.take((size_t)NT * CG * 256);
        r.W3p = c.take((size_t)NT * NG * 256);
        r.g3p = c.take(NT * 32); r.b3p = c.take(NT * 32); r.c3p = c.take(NT * 32);
        r.Wscp = r.sclin ? c.take((size_t)NT * KG * 256) : 0;
    }
    for (auto& l : h->lin) {
        const int NT = cdiv(l.l.N, 32), KG = groups_of(l.l.K);
        l.Wp = c.take((size_t)NT * KG * 256);
        l.bp = c.take(NT * 32);
        l.gp = l.lnact ? c.take((size_t)KG * 8 + 32) : 0;
        l.betap = l.lnact ? c.take((size_t)KG * 8 + 32) : 0;
    }
    h->arena_floats = c.off;
}

int free_workspace(dsg_handle* h) {
    for (int i = 0; i < 2; ++i)
        if (h->gexec[i]) { hipGraphExecDestroy(h->gexec[i]); h->gexec[i] = nullptr; }
    h->g_rows = -1;
    void* ptrs[] = {h->ws, h->condfrag, h->tb, h->st, h->tvals, h->ts_ident, h->eps, h->ywork};
    for (void* p : ptrs)
        if (p) hipFree(p);
    h->ws = h->condfrag = h->tb = h->st = h->tvals = h->eps = h->ywork = nullptr;
    h->ts_ident = nullptr;
    h->cap_rows = h->cap_entries = 0;
    return 0;
}

__global__ void k_iota(int* p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = i;
}
__global__ void k_linspace_t(float* p, int T) {  // t = i / T in float32, as `torch.full(..., i) / T` (MSR.py:126)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x) p[i] = (float)i / (float)T;
}

int ensure_workspace(dsg_handle* h, int rows, int entries) {
    if (rows <= h->cap_rows && entries <= h->cap_entries) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)cs;
    const int nrows = rows > h->cap_rows ? rows : h->cap_rows;
    const int nent = entries > h->cap_entries ? entries : h->cap_entries;
    HIPCK(hipDeviceSynchronize());
    free_workspace(h);
    const size_t tiles = (size_t)cdiv(nrows, 32) * 2;  // two passes
    const int D = h->d.input_dim, CG = groups_of(h->d.cond_dim);
    HIPCK(hipMalloc(&h->ws, tiles * h->per_tile_floats * sizeof(float)));
    HIPCK(hipMemset(h->ws, 0, tiles * h->per_tile_floats * sizeof(float)));
    HIPCK(hipMalloc(&h->condfrag, (tiles / 2) * CG * 256 * sizeof(float)));
    HIPCK(hipMalloc(&h->tb, (size_t)nent * h->tb_stride * sizeof(float)));
    HIPCK(hipMalloc(&h->st, (size_t)nent * h->td * sizeof(float)));
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

struct RunCtx {
    int nrows;           // rows per pass
    int npass;           // 1 or 2
    int uncond_tiles;    // leading tiles without the condition term
    const float* y;      // row-major [nrows][D]
    float* eps_out;      // row-major [npass][nrows][D]
    const int* step_ptr; // or null
    const int* ts;       // or null
};

Seg seg_of(const dsg_handle* h, int tid, size_t cap_tiles) {
    const TensorInfo& t = h->tensors[tid];
    Seg s;
    s.data = h->ws + t.data_off * cap_tiles;
    s.stats = h->ws + t.stats_off * cap_tiles;
    s.groups = groups_of(t.width);
    s.width = t.width;
    return s;
}

void fill_block_args(const dsg_handle* h, const Op& op, const RunCtx& c, BlockArgs& a) {
    const size_t cap_tiles = (size_t)cdiv(h->cap_rows, 32) * 2;
    const int tpp = cdiv(c.nrows, 32);
    const ResP& r = h->res[op.p];
    const float* A = h->arena;
    memset(&a, 0, sizeof a);
    a.in0 = seg_of(h, op.in0, cap_tiles);
    if (op.in1 >= 0) a.in1 = seg_of(h, op.in1, cap_tiles);
    a.W1 = A + r.W1p; a.gamma1 = A + r.g1p; a.beta1 = A + r.b1p;
    a.tbias = h->tb + r.tb_off; a.step_ptr = c.step_ptr; a.ts = c.ts; a.tb_stride = h->tb_stride;
    a.W2 = A + r.W2p; a.gamma2 = A + r.g2p; a.beta2 = A + r.b2p; a.c2 = A + r.c2p;
    a.Wc = A + r.Wcp; a.condfrag = h->condfrag; a.cond_groups = groups_of(h->d.cond_dim);
    a.W3 = A + r.W3p; a.gamma3 = A + r.g3p; a.beta3 = A + r.b3p; a.c3 = A + r.c3p;
    a.Wsc = r.sclin ? A + r.Wscp : nullptr;
    const Seg o = seg_of(h, op.out, cap_tiles);
    a.out = const_cast<float*>(o.data); a.out_stats = const_cast<float*>(o.stats);
    a.ntiles = tpp * c.npass; a.tiles_per_pass = tpp; a.uncond_tiles = c.uncond_tiles; a.nrows = c.nrows;
}

void fill_lin_args(const dsg_handle* h, const Op& op, const RunCtx& c, LinArgs& a) {
    const size_t cap_tiles = (size_t)cdiv(h->cap_rows, 32) * 2;
    const int tpp = cdiv(c.nrows, 32);
    const LinOpP& l = h->lin[op.p];
    const float* A = h->arena;
    memset(&a, 0, sizeof a);
    a.W = A + l.Wp; a.bias = A + l.bp;
    a.in_width = l.l.K; a.in_groups = groups_of(l.l.K);
    a.out_width = l.l.N;
    a.ntiles = tpp * c.npass; a.tiles_per_pass = tpp; a.nrows = c.nrows;
    if (op.kind == OP_PROJ) a.in_rm = c.y;
    else a.in = seg_of(h, op.in0, cap_tiles);
    if (op.kind == OP_FINAL) {
        a.gamma = A + l.gp; a.beta = A + l.betap; a.out_rm = c.eps_out;
    } else {
        const Seg o = seg_of(h, op.out, cap_tiles);
        a.out = const_cast<float*>(o.data); a.out_stats = const_cast<float*>(o.stats);
    }
}

void launch_op(const dsg_handle* h, const Op& op, const RunCtx& c, hipStream_t s) {
    if (op.kind == OP_RES) {
        BlockArgs a;
        fill_block_args(h, op, c, a);
        launch_res(h->res[op.p].N, h->res[op.p].sclin, a, s);
    } else {
        LinArgs a;
        fill_lin_args(h, op, c, a);
        const int inmode = op.kind == OP_PROJ ? IN_ROWMAJOR : IN_FRAG;
        const int outmode = op.kind == OP_FINAL ? OUT_ROWMAJOR : OUT_FRAG;
        launch_lin(h->lin[op.p].l.N, inmode, outmode, op.kind == OP_FINAL, a, s);
    }
}

void run_unet(const dsg_handle* h, const RunCtx& c, hipStream_t s) {
    for (const Op& op : h->ops) launch_op(h, op, c, s);
}

// time path for `entries` t values already in h->tvals
void run_time_path(dsg_handle* h, int entries, hipStream_t s) {
    const int half = h->d.proj_dim / 2, td = h->td;
    const Param* P = h->params.data();
    hipLaunchKernelGGL(k_time_embed, dim3(entries), dim3(256), (2 * half + td) * sizeof(float), s, h->tvals, h->freq, half,
                       P[h->temb_l1w].ptr, P[h->temb_l1b].ptr, P[h->temb_l2w].ptr, P[h->temb_l2b].ptr, td, h->st);
    const int nb = (int)h->res.size();
    hipLaunchKernelGGL(k_time_table, dim3(entries, nb < 8 ? nb : 8), dim3(256), td * sizeof(float), s, h->st, td, h->tdesc_dev,
                       nb, h->tb, h->tb_stride);
}

int check_bound(const dsg_handle* h) {
    if (!h) return fail("null handle");
    if (!h->bound) return fail("dsg_bind_weights has not been called");
    return 0;
}

}  // namespace

// ======================================================================================================
extern "C" {

const char* dsg_last_error(void) { return g_err.c_str(); }

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
    int cur = add_tensor(h, d.proj_dim);
    h->ops.push_back(Op{OP_PROJ, 0, -1, -1, cur, "feature_proj"});
    skips.push_back(cur);
    int w = d.proj_dim, idx = 0;
    char nm[64];
    auto push_down_res = [&](int width) {
        snprintf(nm, sizeof nm, "down.%d.res", idx);
        const int p = add_res(h, nm, width, 0, width);
        const int out = add_tensor(h, width);
        h->ops.push_back(Op{OP_RES, p, cur, -1, out, nm});
        cur = out; skips.push_back(cur); ++idx;
    };
    for (int i = 0; i < d.n_res; ++i) {
        for (int b = 0; b < d.n_blocks; ++b) push_down_res(w);
        snprintf(nm, sizeof nm, "down.%d.lin", idx);
        LinOpP l; l.l = add_linear(h, nm, w, d.dims[i]);
        h->lin.push_back(l);
        const int out = add_tensor(h, d.dims[i]);
        h->ops.push_back(Op{OP_LIN, (int)h->lin.size() - 1, cur, -1, out, nm});
        cur = out; skips.push_back(cur); ++idx;
        w = d.dims[i];
        if (i == d.n_res - 1)
            for (int b = 0; b < d.n_blocks; ++b) push_down_res(w);
    }
    for (int m = 1; m <= 2; ++m) {
        snprintf(nm, sizeof nm, "middle.res%d", m);
        const int p = add_res(h, nm, w, 0, w);
        const int out = add_tensor(h, w);
        h->ops.push_back(Op{OP_RES, p, cur, -1, out, nm});
        cur = out;
    }
    idx = 0;
    auto push_up_res = [&](int width) {
        snprintf(nm, sizeof nm, "up.%d.res", idx);
        const int sk = skips.back(); skips.pop_back();
        const int p = add_res(h, nm, width, h->tensors[sk].width, width);
        const int out = add_tensor(h, width);
        h->ops.push_back(Op{OP_RES, p, cur, sk, out, nm});
        cur = out; ++idx;
    };
    for (int i = d.n_res - 1; i >= 0; --i) {
        for (int b = 0; b < d.n_blocks + 1; ++b) push_up_res(w);
        const int nw = i > 0 ? d.dims[i - 1] : d.proj_dim;
        snprintf(nm, sizeof nm, "up.%d.lin", idx);
        LinOpP l; l.l = add_linear(h, nm, w, nw);
        h->lin.push_back(l);
        const int out = add_tensor(h, nw);
        h->ops.push_back(Op{OP_LIN, (int)h->lin.size() - 1, cur, -1, out, nm});
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
        if (r.in1 && r.in1 != r.in0) { fail("internal: skip width mismatch"); delete h; return nullptr; }
    carve(h);
    bool ok = hipMalloc(&h->arena, h->arena_floats * sizeof(float)) == hipSuccess &&
              hipMemset(h->arena, 0, h->arena_floats * sizeof(float)) == hipSuccess &&
              hipMalloc(&h->tdesc_dev, h->res.size() * sizeof(TimeBlockDesc)) == hipSuccess &&
              hipMalloc(&h->freq, (d.proj_dim / 2) * sizeof(float)) == hipSuccess &&
              hipMalloc(&h->red, 2 * kRedBlocks * sizeof(double)) == hipSuccess &&
              hipMalloc(&h->step_dev, 64) == hipSuccess &&
              hipMalloc(&h->call_dev, sizeof(CallParams)) == hipSuccess &&
              hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) == hipSuccess;
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
    if (!h) return;
    hipDeviceSynchronize();
    free_workspace(h);
    if (h->arena) hipFree(h->arena);
    if (h->tdesc_dev) hipFree(h->tdesc_dev);
    if (h->freq) hipFree(h->freq);
    if (h->red) hipFree(h->red);
    if (h->step_dev) hipFree(h->step_dev);
    if (h->call_dev) hipFree(h->call_dev);
    if (h->cap_stream) hipStreamDestroy(h->cap_stream);
    delete h;
}

int dsg_param_count(const dsg_handle* h) { return h ? (int)h->params.size() : 0; }
const char* dsg_param_name(const dsg_handle* h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : "";
}
long long dsg_param_numel(const dsg_handle* h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].numel : -1;
}

int dsg_bind_weights(dsg_handle* h, const float* const* ptrs, int n, void* stream) {
    if (!h) return fail("null handle");
    if (n != (int)h->params.size()) return fail("dsg_bind_weights: got %d pointers, the model has %d tensors", n, (int)h->params.size());
    for (int i = 0; i < n; ++i) {
        if (!ptrs[i]) return fail("dsg_bind_weights: null pointer for %s", h->params[i].name.c_str());
        h->params[i].ptr = ptrs[i];
    }
    hipStream_t s = (hipStream_t)stream;
    const Param* P = h->params.data();
    float* A = h->arena;
    auto pack = [&](const LinearP& l, int w0, int w1, size_t off) {
        const int NT = cdiv(l.N, 32);
        const size_t total = (size_t)NT * (groups_of(w0) + groups_of(w1)) * 256;
        hipLaunchKernelGGL(k_pack_linear, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s,
                           P[l.w].ptr, l.N, l.K, w0, w1, A + off, NT);
    };
    auto padv = [&](const float* a, const float* b, int w0, int w1, size_t off, int npad) {
        hipLaunchKernelGGL(k_pad_vec, dim3(cdiv(npad, 256)), dim3(256), 0, s, a, b, w0, w1, A + off, npad);
    };
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
        if (r.sclin) pack(r.sc, r.in0, r.in1, r.Wscp);
        td[i] = TimeBlockDesc{P[r.te.w].ptr, P[r.te.b].ptr, P[r.l1.b].ptr, r.N, r.tb_off};
    }
    for (const LinOpP& l : h->lin) {
        const int NT = cdiv(l.l.N, 32), KG = groups_of(l.l.K);
        pack(l.l, l.l.K, 0, l.Wp);
        padv(P[l.l.b].ptr, nullptr, l.l.N, 0, l.bp, NT * 32);
        if (l.lnact) {
            padv(P[l.ln.w].ptr, nullptr, l.l.K, 0, l.gp, KG * 8 + 32);
            padv(P[l.ln.b].ptr, nullptr, l.l.K, 0, l.betap, KG * 8 + 32);
        }
    }
    HIPCK(hipMemcpyAsync(h->tdesc_dev, td.data(), td.size() * sizeof(TimeBlockDesc), hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));  // td is a host temporary
    HIPCK(hipGetLastError());
    h->bound = true;
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
    RunCtx c{B, 1, 0, x, out, nullptr, h->ts_ident};
    run_unet(h, c, s);
    HIPCK(hipGetLastError());
    return 0;
}

static int enqueue_step(dsg_handle* h, const RunCtx& c, const UpdateArgs& u, bool renorm, hipStream_t s,
                        hipEvent_t* ev = nullptr) {
    if (ev) {  // DSG_SAMPLE_PROFILE: one event pair per operator launch
        for (size_t i = 0; i < h->ops.size(); ++i) {
            HIPCK(hipEventRecord(ev[2 * i], s));
            launch_op(h, h->ops[i], c, s);
            HIPCK(hipEventRecord(ev[2 * i + 1], s));
        }
    } else {
        run_unet(h, c, s);
    }
    const unsigned ublocks = (unsigned)(((u.n + 3) / 4 + 255) / 256 < 2048 ? ((u.n + 3) / 4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_update, dim3(ublocks), dim3(256), 0, s, u);
    if (renorm) {
        hipLaunchKernelGGL(k_renorm_sum, dim3(kRedBlocks), dim3(256), 0, s, u.y, u.n, h->red);
        hipLaunchKernelGGL(k_renorm_sqdiff, dim3(kRedBlocks), dim3(256), 0, s, u.y, u.n, h->red, h->red + kRedBlocks);
        hipLaunchKernelGGL(k_renorm_apply, dim3(kRedBlocks), dim3(256), 0, s, u.y, u.n, h->red, h->red + kRedBlocks);
    }
    hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, s, h->step_dev);
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

int dsg_sample(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed, float omega,
               const float* coef, int T, float* out, int B, int flags, void* stream) {
    if (check_bound(h)) return 1;
    if (B < 1 || T < 1) return fail("B and T must be >= 1");
    if (!coef || !cond || !out) return fail("dsg_sample: null pointer argument");
    if (ensure_workspace(h, B, T)) return 1;
    hipStream_t s = (hipStream_t)stream;
    const int D = h->d.input_dim, CG = groups_of(h->d.cond_dim), tpp = cdiv(B, 32);
    const size_t n = (size_t)B * D;

    // per-call setup: time table for all T steps, condition fragments, start state, step counter
    hipLaunchKernelGGL(k_linspace_t, dim3(cdiv(T, 256)), dim3(256), 0, s, h->tvals, T);
    run_time_path(h, T, s);
    hipLaunchKernelGGL(k_cond_frag, dim3(cdiv(tpp * CG * 256, 256)), dim3(256), 0, s, cond, (const float*)nullptr, B,
                       h->d.cond_dim, CG, h->condfrag, tpp);
    if (y_T) HIPCK(hipMemcpyAsync(h->ywork, y_T, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    else hipLaunchKernelGGL(k_randn, dim3(2048), dim3(256), 0, s, h->ywork, n, seed, 0xFFFFFFFFu);
    const int start = T - 1;
    HIPCK(hipMemcpyAsync(h->step_dev, &start, sizeof(int), hipMemcpyHostToDevice, s));

    const CallParams cp{noise, coef, omega, T, seed};
    HIPCK(hipMemcpyAsync(h->call_dev, &cp, sizeof cp, hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));  // `start` and `cp` are host temporaries

    RunCtx c{B, 2, tpp, h->ywork, h->eps, h->step_dev, nullptr};
    UpdateArgs u;
    u.eps = h->eps; u.y = h->ywork; u.cp = h->call_dev; u.step_ptr = h->step_dev; u.n = n;

    const int n_renorm = T < 4 ? T : 4;  // steps i > T-5 (MSR.py:136)
    if (flags & DSG_SAMPLE_PROFILE) {
        const size_t nops = h->ops.size();
        h->op_ms.assign(nops, 0.0);
        h->op_calls.assign(nops, 0);
        std::vector<hipEvent_t> ev(2 * nops);
        for (auto& e : ev) HIPCK(hipEventCreate(&e));
        int rc = 0;
        for (int k = 0; k < T && !rc; ++k) rc = enqueue_step(h, c, u, k < n_renorm, s, ev.data());
        for (auto& e : ev) hipEventDestroy(e);
        if (rc) return 1;
    } else if (flags & DSG_SAMPLE_NO_GRAPH) {
        for (int k = 0; k < T; ++k)
            if (enqueue_step(h, c, u, k < n_renorm, s)) return 1;
    } else {
        // the per-step graph depends only on the workspace (batch size); everything per call is in device memory
        if (!(h->gexec[0] && h->g_rows == B)) {
            for (int i = 0; i < 2; ++i)
                if (h->gexec[i]) { hipGraphExecDestroy(h->gexec[i]); h->gexec[i] = nullptr; }
            for (int variant = 0; variant < 2; ++variant) {  // 0: with renorm, 1: without
                hipGraph_t g = nullptr;
                HIPCK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
                const int rc = enqueue_step(h, c, u, variant == 0, h->cap_stream);
                hipError_t e = hipStreamEndCapture(h->cap_stream, &g);
                if (rc) return 1;
                if (e != hipSuccess) return fail("hipStreamEndCapture: %s", hipGetErrorString(e));
                e = hipGraphInstantiate(&h->gexec[variant], g, nullptr, nullptr, 0);
                hipGraphDestroy(g);
                if (e != hipSuccess) return fail("hipGraphInstantiate: %s", hipGetErrorString(e));
            }
            h->g_rows = B;
        }
        for (int k = 0; k < T; ++k) HIPCK(hipGraphLaunch(h->gexec[k < n_renorm ? 0 : 1], s));
    }
    HIPCK(hipMemcpyAsync(out, h->ywork, n * sizeof(float), hipMemcpyDeviceToDevice, s));
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

int dsg_op_count(const dsg_handle* h) { return h ? (int)h->ops.size() : 0; }

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
    if (flops_per_row) *flops_per_row = 2.0 * (2.0 * macs2 + macs_cond);
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
    RunCtx c{B, 2, cdiv(B, 32), h->ywork, h->eps, h->step_dev, nullptr};
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
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms_avg = ms / iters;
    return 0;
}

}  // extern "C"
