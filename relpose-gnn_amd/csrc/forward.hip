// Composite forwards: the whole ResNet (BasicBlock) encoder and the whole GNN + heads, as sequences of the
// kernels in gemm_f32.hip / encoder_ops.hip / gnn_ops.hip on one stream.  No allocation, no synchronisation:
// the caller passes a workspace sized by rpg_*_workspace_bytes().
//
// Reference control flow restated here:
//   encoder  torchvision 0.9.1 ResNet._forward_impl with BasicBlock (call site posenet.py:1037)
//   GNN      /root/reference/python/niantic/modules/posenet.py:1052-1091 and my_gnn_layer.py:293-311
#include "rpg_common.h"

namespace rpg {
int launch_relu_inplace(float* x, long n_floats, hipStream_t s);
bool stem_pool_supported(int h, int w, int cout);
int launch_stem_pool(const float* x_nchw, const float* wpack, const float* shift, float* out, int n, int h, int w, hipStream_t s);
}

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

struct Carver {
    char* base;
    size_t off;
    template <class T>
    T* take(size_t count) {
        T* p = reinterpret_cast<T*>(base + off);
        off += align_up(count * sizeof(T), 256) + rpg::kWorkspaceSkew;
        return p;
    }
};

// Buffer plan of the encoder (floats): NHWC4 input, stem output, 4 rotating block buffers, pooled vector.
struct ResnetPlan {
    int h1, w1, h2, w2;
    size_t in4, stem, blk, pool, total_bytes;
};

ResnetPlan plan_resnet(int n, int h, int w, const int* planes) {
    ResnetPlan p{};
    p.h1 = conv_out(h, 7, 2, 3); p.w1 = conv_out(w, 7, 2, 3);
    p.h2 = conv_out(p.h1, 3, 2, 1); p.w2 = conv_out(p.w1, 3, 2, 1);
    p.in4 = (size_t)n * h * w * 4;
    p.stem = (size_t)n * p.h1 * p.w1 * planes[0];
    size_t blk = 0;
    int hh = p.h2, ww = p.w2;
    for (int l = 0; l < 4; ++l) {
        if (l > 0) { hh = conv_out(hh, 3, 2, 1); ww = conv_out(ww, 3, 2, 1); }
        const size_t sz = (size_t)n * hh * ww * planes[l];
        if (sz > blk) blk = sz;
    }
    p.blk = blk;
    p.pool = (size_t)n * planes[3];
    p.total_bytes = align_up(p.in4 * 4, 256) + align_up(p.stem * 4, 256) + 4 * align_up(p.blk * 4, 256) +
                    align_up(p.pool * 4, 256) + align_up(rpg::split_scratch_bytes(), 256) + 8 * rpg::kWorkspaceSkew;
    return p;
}

}  // namespace

extern "C" size_t rpg_resnet_workspace_bytes(int n, int h, int w, const int* planes) {
    if (n <= 0 || h <= 0 || w <= 0 || !planes) return 0;
    return plan_resnet(n, h, w, planes).total_bytes;
}

extern "C" int rpg_resnet_forward_f32(const float* const* tensors, int n_tensors, const int* blocks, const int* planes,
                                      int feat_dim, const float* x_nchw, int n, int h, int w, float* feat,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!tensors || !blocks || !planes || !x_nchw || !feat || !workspace || n <= 0 || h <= 0 || w <= 0 || feat_dim <= 0)
        return RPG_ERR_BAD_ARG;
    // tensor count: stem 4 + per block 8 (+4 with downsample) + fc 2; every 4th entry (u_wino43) may be NULL
    int expect = 4 + 2, cin = planes[0];
    for (int l = 0; l < 4; ++l)
        for (int b = 0; b < blocks[l]; ++b) {
            const int stride = (l > 0 && b == 0) ? 2 : 1;
            expect += 8 + ((stride != 1 || cin != planes[l]) ? 4 : 0);
            cin = planes[l];
        }
    if (n_tensors != expect) return RPG_ERR_BAD_ARG;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i] && !((i & 3) == 3 && i < n_tensors - 2)) return RPG_ERR_BAD_ARG;
    const ResnetPlan p = plan_resnet(n, h, w, planes);
    if (workspace_bytes < p.total_bytes) return RPG_ERR_WORKSPACE;
    hipStream_t s = rpg::as_stream(stream);

    Carver cv{reinterpret_cast<char*>(workspace), 0};
    float* in4 = cv.take<float>(p.in4);
    float* stem = cv.take<float>(p.stem);
    float* buf[4];
    for (int i = 0; i < 4; ++i) buf[i] = cv.take<float>(p.blk);
    float* pool = cv.take<float>(p.pool);
    const size_t scratch_bytes = rpg::split_scratch_bytes();
    rpg::ScratchScope scratch(cv.take<char>(scratch_bytes), scratch_bytes, s);  // split-K partial tiles of this call

    int rc;
    int ti = 0;
    if (tensors[3] && rpg::stem_pool_supported(h, w, planes[0])) {
        // the fused stem: NCHW input -> conv7x7/2 + BN + ReLU + maxpool3x3/2 -> pooled NHWC, one kernel (csrc/stem.hip);
        // tensors[3] = its packed weight operands (BN scale folded in), tensors[2] = the BN shift
        if ((rc = rpg::launch_stem_pool(x_nchw, tensors[3], tensors[2], buf[0], n, h, w, s)) != RPG_OK) return rc;
    } else {
        if ((rc = rpg_nchw3_to_nhwc4_f32(x_nchw, in4, n, h, w, stream)) != RPG_OK) return rc;
        // stem: conv7x7/2 pad 3 (3 -> planes[0], input channels padded to 4 with zero weights) + BN + ReLU
        if ((rc = rpg::launch_conv(in4, tensors[ti], tensors[ti + 1], tensors[ti + 2], nullptr, stem, n, h, w, 4,
                                   planes[0], 7, 7, 2, 3, 1, s, 3)) != RPG_OK)
            return rc;
        if ((rc = rpg_maxpool3x3s2_nhwc_f32(stem, buf[0], n, p.h1, p.w1, planes[0], stream)) != RPG_OK) return rc;
    }
    ti += 4;

    int cur = 0, hh = p.h2, ww = p.w2;
    cin = planes[0];
    for (int l = 0; l < 4; ++l) {
        for (int b = 0; b < blocks[l]; ++b) {
            const int stride = (l > 0 && b == 0) ? 2 : 1;
            const int c = planes[l];
            const bool ds = (stride != 1 || cin != c);
            const int ho = conv_out(hh, 3, stride, 1), wo = conv_out(ww, 3, stride, 1);
            float* X = buf[cur];
            float* T = buf[(cur + 1) & 3];
            float* Y = buf[(cur + 2) & 3];
            float* D = buf[(cur + 3) & 3];
            // conv1 3x3/stride + BN + ReLU
            // Winograd where it pays (rpg::wino_pays: addressable with 32-bit buffer offsets and enough workgroups)
            const bool wino1 = stride == 1 && tensors[ti + 3] && rpg::wino_pays(n, hh, ww, cin, c);
            const bool wino2 = tensors[ti + 7] && rpg::wino_pays(n, ho, wo, c, c);
            if (wino1)
                rc = rpg::launch_conv_wino(X, tensors[ti + 3], tensors[ti + 1], tensors[ti + 2], nullptr, T, n, hh, ww, cin,
                                           c, 1, s);
            else
                rc = rpg::launch_conv(X, tensors[ti], tensors[ti + 1], tensors[ti + 2], nullptr, T, n, hh, ww, cin, c, 3, 3,
                                      stride, 1, 1, s);
            if (rc != RPG_OK) return rc;
            const float* identity = X;
            if (ds) {   // downsample: conv1x1/stride + BN (no activation)
                if ((rc = rpg::launch_conv(X, tensors[ti + 8], tensors[ti + 9], tensors[ti + 10], nullptr, D, n, hh, ww,
                                           cin, c, 1, 1, stride, 0, 0, s)) != RPG_OK)
                    return rc;
                identity = D;
            }
            // conv2 3x3/1 + BN + identity + ReLU
            if (wino2)
                rc = rpg::launch_conv_wino(T, tensors[ti + 7], tensors[ti + 5], tensors[ti + 6], identity, Y, n, ho, wo, c, c,
                                           1, s);
            else
                rc = rpg::launch_conv(T, tensors[ti + 4], tensors[ti + 5], tensors[ti + 6], identity, Y, n, ho, wo, c, c, 3,
                                      3, 1, 1, 1, s);
            if (rc != RPG_OK) return rc;
            ti += ds ? 12 : 8;
            cur = (cur + 2) & 3;
            hh = ho; ww = wo; cin = c;
        }
    }
    if ((rc = rpg_global_avgpool_nhwc_f32(buf[cur], pool, n, hh * ww, cin, stream)) != RPG_OK) return rc;
    rpg::GatherSrc src{};
    src.n = 1; src.a[0] = pool; src.idx[0] = nullptr; src.ld[0] = cin; src.width[0] = cin;
    return rpg::launch_linear(src, tensors[ti], tensors[ti + 1], nullptr, feat, n, feat_dim, 0, s);
}

// ------------------------------------------------------------------------------------------------------------
// GNN + heads
// ------------------------------------------------------------------------------------------------------------
namespace {
enum GnnTensor {
    T_PROJ_W, T_PROJ_B, T_EDGE0_W, T_EDGE0_B, T_EDGE2_W, T_EDGE2_B, T_MSG0_W, T_MSG0_B, T_MSG2_W, T_MSG2_B,
    T_GTP_W, T_GTP_B, T_ATTW_W, T_ATTW_B, T_UPD0_W, T_UPD0_B, T_UPD2_W, T_UPD2_B, T_HEADN_W, T_HEADN_B,
    T_HEADE_W, T_HEADE_B, T_COUNT,
    // optional node/edge split of the concatenated-input Linears (enables the per-node precompute path):
    T_PROJN_W = T_COUNT,   // [2D][D]  = rows cat(proj_edge.W[:, :D], proj_edge.W[:, D:])
    T_NODE3_W,             // [3D][D]  = rows cat(edge_mlp.0.W[:, :D], edge_mlp.0.W[:, D:2D], mlp.0.W[:, :D])
    T_EDGE0E_W,            // [D][D]   = edge_mlp.0.W[:, 2D:]
    T_MSG0E_W,             // [D][D]   = mlp.0.W[:, D:]
    T_COUNT_SPLIT
};

struct GnnPlan {
    size_t total_bytes;
};
size_t gnn_bytes(int n, int e, int d) {
    const size_t c = d / 8;
    size_t b = 0;
    auto add = [&](size_t bytes, int count = 1) { b += count * (align_up(bytes, 256) + rpg::kWorkspaceSkew); };
    add((size_t)4 * e * 8);                 // ends
    add((size_t)(n + 1) * 4);               // rowptr
    add((size_t)n * 4);                     // cursor
    add((size_t)e * 4);                     // perm
    add((size_t)e * d * 4, 6);              // e0, e1, raw edge update, hidden, msg, att (fused aggregation: mean messages)
    add((size_t)e * 3 * c * 4);             // g|theta|phi
    add((size_t)e * c * 4);                 // y
    add((size_t)n * d * 4, 4);              // agg, node hidden, x ping-pong
    add((size_t)n * 3 * d * 4);             // per-node partial products of the split Linears
    add((size_t)(e > 2 * n ? e : 2 * n) * d * 2);   // bf16 image of a Linear's input (bf16 GNN only)
    add(rpg::split_scratch_bytes());                // stream-K partial tiles
    return b;
}
}  // namespace

extern "C" size_t rpg_gnn_workspace_bytes(int n, int e, int d) {
    if (n <= 0 || e <= 0 || d <= 0 || (d & 31)) return 0;
    return gnn_bytes(n, e, d);
}

namespace {
// wb: null (every Linear in fp32), or the bf16 images of the 10 GEMM weights of the split formulation, in the order
// of BfWeight (then the Linears run on v_mfma_f32_32x32x16_bf16 with bf16 inputs / weights, fp32 accumulation, fp32
// bias / residual / output; everything that is not a GEMM stays fp32)
enum BfWeight { B_PROJN, B_NODE3, B_EDGE0E, B_EDGE2, B_MSG0E, B_MSG2, B_GTP, B_ATTW, B_UPD0, B_UPD2, B_COUNT };

int gnn_forward_impl(const float* const* tensors, int n_tensors, const void* const* wb, const float* feat, const int64_t* esrc,
                     const int64_t* edst, int64_t node_offset, int n, int e, int d, int gnn_recursion, float* abs_pose,
                     float* rel_pose, float* node_out, float* edge_out, int32_t* status, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!tensors || (n_tensors != T_COUNT && n_tensors != T_COUNT_SPLIT) || !feat || !esrc || !edst || !abs_pose || !rel_pose || !status || !workspace ||
        n <= 0 || e <= 0 || d <= 0 || (d & 31) || gnn_recursion < 0)
        return RPG_ERR_BAD_ARG;
    for (int i = 0; i < n_tensors; ++i)
        if (!tensors[i]) return RPG_ERR_BAD_ARG;
    if (workspace_bytes < gnn_bytes(n, e, d)) return RPG_ERR_WORKSPACE;
    const bool split = (n_tensors == T_COUNT_SPLIT) && (wb || rpg::gnn_split_enabled());
    if (wb && !split) return RPG_ERR_BAD_ARG;            // the bf16 Linears exist for the split formulation only
    hipStream_t s = rpg::as_stream(stream);
    const int c = d / 8;

    Carver cv{reinterpret_cast<char*>(workspace), 0};
    int64_t* ends = cv.take<int64_t>((size_t)4 * e);
    int32_t* rowptr = cv.take<int32_t>((size_t)n + 1);
    int32_t* cursor = cv.take<int32_t>((size_t)n);
    int32_t* perm = cv.take<int32_t>((size_t)e);
    float* ebuf[2] = {cv.take<float>((size_t)e * d), cv.take<float>((size_t)e * d)};
    float* eraw = cv.take<float>((size_t)e * d);
    float* hid = cv.take<float>((size_t)e * d);
    float* msg = cv.take<float>((size_t)e * d);
    float* att = cv.take<float>((size_t)e * d);
    float* gtp = cv.take<float>((size_t)e * 3 * c);
    float* yat = cv.take<float>((size_t)e * c);
    float* agg = cv.take<float>((size_t)n * d);
    float* nhid = cv.take<float>((size_t)n * d);
    float* xbuf[2] = {cv.take<float>((size_t)n * d), cv.take<float>((size_t)n * d)};
    float* node3 = cv.take<float>((size_t)n * 3 * d);
    void* abf = cv.take<unsigned short>((size_t)(e > 2 * n ? e : 2 * n) * d);
    const size_t scratch_bytes = rpg::split_scratch_bytes();
    rpg::ScratchScope scratch(cv.take<char>(scratch_bytes), scratch_bytes, s);

    int rc;
    if ((rc = rpg_graph_prepare(esrc, edst, node_offset, e, n, ends, rowptr, cursor, perm, status, stream)) != RPG_OK) return rc;
    const int64_t* src = ends;
    const int64_t* dst = ends + e;
    const int64_t* lo = ends + 2 * (size_t)e;
    const int64_t* hi = ends + 3 * (size_t)e;

    auto linear = [&](int ns, const float* a0, const int64_t* i0, int w0, const float* a1, const int64_t* i1, int w1,
                      const float* a2, const int64_t* i2, int w2, int wt, const float* residual, float* out, int m,
                      int n_out, int relu, float* out_relu = nullptr, bool with_bias = true) {
        const float* bias = with_bias ? tensors[wt + 1] : nullptr;
        rpg::GatherSrc g{};
        g.n = ns;
        g.a[0] = a0; g.idx[0] = i0; g.ld[0] = w0; g.width[0] = w0;
        g.a[1] = a1; g.idx[1] = i1; g.ld[1] = w1; g.width[1] = w1;
        g.a[2] = a2; g.idx[2] = i2; g.ld[2] = w2; g.width[2] = w2;
        for (int i = 0; i < 3; ++i) g.rows[i] = g.idx[i] ? (n > e ? n : e) : 0;       // node / edge ids index n- or e-row arrays
        if (wb) {            // bf16: only ungathered sources reach here (split formulation); concatenate them in bf16
            int bw = -1;
            switch (wt) {
                case T_EDGE2_W: bw = B_EDGE2; break;
                case T_MSG2_W: bw = B_MSG2; break;
                case T_GTP_W: bw = B_GTP; break;
                case T_ATTW_W: bw = B_ATTW; break;
                case T_UPD0_W: bw = B_UPD0; break;
                case T_UPD2_W: bw = B_UPD2; break;
                default: return RPG_ERR_BAD_ARG;
            }
            const int k = w0 + (ns > 1 ? w1 : 0);
            if (ns > 2 || i0 || i1) return RPG_ERR_BAD_ARG;
            int r2;
            if ((r2 = rpg::launch_f32_to_bf16(a0, w0, abf, k, 0, m, w0, s)) != RPG_OK) return r2;
            if (ns > 1 && (r2 = rpg::launch_f32_to_bf16(a1, w1, abf, k, w0, m, w1, s)) != RPG_OK) return r2;
            if (out_relu) return RPG_ERR_BAD_ARG;          // the bf16 Linears have no dual-store epilogue
            return rpg::launch_linear_bf16(abf, wb[bw], bias, residual, nullptr, nullptr, nullptr, n_out, out, m, k,
                                           n_out, relu, s);
        }
        return rpg::launch_linear(g, tensors[wt], bias, residual, out, m, n_out, relu, s, nullptr, out_relu);
    };

    // A Linear fed by cat[x[a], x[b], e] splits as W_a x[a] + W_b x[b] + W_e e: the node terms are computed once per
    // NODE (n rows) and added as gathered rows in the edge GEMM's epilogue (summation order changes only).
    auto node_gemm = [&](const float* xin, int wt, int n_out) {
        rpg::GatherSrc g{};
        g.n = 1; g.a[0] = xin; g.idx[0] = nullptr; g.ld[0] = d; g.width[0] = d;
        if (wb) {
            int r2;
            if ((r2 = rpg::launch_f32_to_bf16(xin, d, abf, d, 0, n, d, s)) != RPG_OK) return r2;
            return rpg::launch_linear_bf16(abf, wb[wt == T_PROJN_W ? B_PROJN : B_NODE3], nullptr, nullptr, nullptr, nullptr,
                                           nullptr, 0, node3, n, d, n_out, 0, s);
        }
        return rpg::launch_linear(g, tensors[wt], nullptr, nullptr, node3, n, n_out, 0, s);
    };
    auto edge_gemm = [&](const float* ein, int wt, int bias_t, const float* r1, const int64_t* i1, const float* r2,
                         const int64_t* i2, float* out) {
        rpg::GatherSrc g{};
        g.n = 1; g.a[0] = ein; g.idx[0] = nullptr; g.ld[0] = d; g.width[0] = d;
        const rpg::GatherRes gr{r1, i1, r2, i2, 3 * d};
        if (wb) {
            int rcb;
            if ((rcb = rpg::launch_f32_to_bf16(ein, d, abf, d, 0, e, d, s)) != RPG_OK) return rcb;
            return rpg::launch_linear_bf16(abf, wb[wt == T_EDGE0E_W ? B_EDGE0E : B_MSG0E], tensors[bias_t], r1, i1, r2, i2, 3 * d,
                                           out, e, d, d, 1, s);
        }
        return rpg::launch_linear(g, tensors[wt], tensors[bias_t], nullptr, out, e, d, 1, s, &gr);
    };

    const bool fuse_agg = rpg::gnn_fuse_agg_enabled();
    if (wb && fuse_agg && split) {
        // ---- bf16 Linears, round 3: every GEMM hands its result to the next GEMM in bf16 FROM ITS EPILOGUE (EpiB::out2 /
        // a bf16 primary output); the 21 separate f32 -> bf16 passes and the 2 in-place ReLU passes per forward of round 2
        // are down to 4 conversions (encoder features, proj_edge output, 2 x the n x D/8 attention vector).  The bf16
        // tensors live in the fp32 buffers this mode does not use (raw edge update, hidden, node hidden, agg).
        typedef unsigned short bf;
        {   // the bf16 tensors below are ALIASED onto fp32 buffers carved above: check that every one fits the buffer it is
            // placed in (all four are exact fits today; a change to gnn_bytes() / the carving order must not silently overlap)
            const size_t e_buf = (size_t)e * d * sizeof(float), n_buf = (size_t)n * d * sizeof(float);
            const size_t a_buf = (size_t)(e > 2 * n ? e : 2 * n) * d * sizeof(bf);
            if (2 * (size_t)e * d * sizeof(bf) > e_buf ||          // eb | enb in eraw, hb | mb in hid
                (size_t)n * 2 * d * sizeof(bf) > n_buf ||          // xab [n][2d] in nhid
                (size_t)n * d * sizeof(bf) > n_buf ||              // nhb [n][d] in agg
                (size_t)n * c * sizeof(bf) > a_buf)                // yb [n][c] in abf
                return RPG_ERR_WORKSPACE;
        }
        bf* eb = reinterpret_cast<bf*>(eraw);                 // [e][d]   current edge features (A of edge_mlp.0's edge block)
        bf* enb = eb + (size_t)e * d;                         // [e][d]   raw edge update (A of mlp.0's edge block)
        bf* hb = reinterpret_cast<bf*>(hid);                  // [e][d]   hidden activations of edge_mlp / mlp
        bf* mb = hb + (size_t)e * d;                          // [e][d]   messages (A of g|theta|phi)
        bf* xab = reinterpret_cast<bf*>(nhid);                // [n][2d]  x | aggregated messages (A of mlp_updating.0; x alone: lda = 2d)
        bf* nhb = reinterpret_cast<bf*>(agg);                 // [n][d]   hidden activations of mlp_updating
        bf* yb = reinterpret_cast<bf*>(abf);                  // [n][c]   attention vector
        auto gemm = [&](const void* a, int lda, int k, int bw, const float* bias, const float* r1, const int64_t* i1, const float* r2,
                        const int64_t* i2, int ldr, void* out, int out_f32, void* out2, int ld2, int relu2, int m, int n_out, int relu) {
            rpg::LinearBf16Out o{};
            o.out = out; o.out_f32 = out_f32; o.out2 = out2; o.ld2 = ld2; o.relu2 = relu2;
            return rpg::launch_linear_bf16_ex(a, lda, wb[bw], bias, r1, i1, r2, i2, ldr, o, m, k, n_out, relu, s);
        };
        if ((rc = rpg::launch_f32_to_bf16(feat, d, xab, 2 * d, 0, n, d, s)) != RPG_OK) return rc;
        // edge_feat = relu(proj_edge(cat[x[min], x[max]]))                               posenet.py:1053-1055
        if ((rc = gemm(xab, 2 * d, d, B_PROJN, nullptr, nullptr, nullptr, nullptr, nullptr, 0, node3, 1, nullptr, 0, 0, n, 2 * d, 0)) != RPG_OK) return rc;
        float* ecur = ebuf[0];
        if ((rc = rpg::launch_gather_add2_relu(node3, lo, hi, tensors[T_PROJ_B], ecur, e, d, s)) != RPG_OK) return rc;
        if ((rc = rpg::launch_f32_to_bf16(ecur, d, eb, d, 0, e, d, s)) != RPG_OK) return rc;
        const float* x = feat;
        for (int r = 0; r < gnn_recursion; ++r) {                                       // posenet.py:1061-1069
            const bool last = r + 1 == gnn_recursion;
            float* xnew = xbuf[r & 1];
            // edge update                                                              my_gnn_layer.py:296-297
            if ((rc = gemm(xab, 2 * d, d, B_NODE3, nullptr, nullptr, nullptr, nullptr, nullptr, 0, node3, 1, nullptr, 0, 0, n, 3 * d, 0)) != RPG_OK) return rc;
            if ((rc = gemm(eb, d, d, B_EDGE0E, tensors[T_EDGE0_B], node3, src, node3 + d, dst, 3 * d, hb, 0, nullptr, 0, 0, e, d, 1)) != RPG_OK) return rc;
            // raw update -> enb (consumed by the message MLP); relu(update) (posenet.py:1065) -> the next recursion's bf16 edge
            // features, or on the last recursion the fp32 ones the heads read
            if (last) rc = gemm(hb, d, d, B_EDGE2, tensors[T_EDGE2_B], nullptr, nullptr, nullptr, nullptr, 0, ebuf[1], 1, enb, d, 0, e, d, 1);
            else rc = gemm(hb, d, d, B_EDGE2, tensors[T_EDGE2_B], nullptr, nullptr, nullptr, nullptr, 0, eb, 0, enb, d, 0, e, d, 1);
            if (rc != RPG_OK) return rc;
            if (last) ecur = ebuf[1];
            // message MLP, attention, aggregation                                      my_gnn_layer.py:301,304-307
            if ((rc = gemm(enb, d, d, B_MSG0E, tensors[T_MSG0_B], node3 + 2 * d, src, nullptr, nullptr, 3 * d, hb, 0, nullptr, 0, 0, e, d, 1)) != RPG_OK) return rc;
            if ((rc = gemm(hb, d, d, B_MSG2, tensors[T_MSG2_B], nullptr, nullptr, nullptr, nullptr, 0, msg, 1, mb, d, 0, e, d, 0)) != RPG_OK) return rc;
            if ((rc = gemm(mb, d, d, B_GTP, tensors[T_GTP_B], nullptr, nullptr, nullptr, nullptr, 0, gtp, 1, nullptr, 0, 0, e, 3 * c, 0)) != RPG_OK) return rc;
            if ((rc = rpg_attention_aggregate_f32(gtp, msg, rowptr, perm, tensors[T_ATTW_B], n, e, c, d, yat, att, stream)) != RPG_OK) return rc;
            if ((rc = rpg::launch_f32_to_bf16(yat, c, yb, c, 0, n, c, s)) != RPG_OK) return rc;
            // att.W on node rows; the aggregate goes straight to the right half of mlp_updating.0's bf16 input
            if ((rc = gemm(yb, c, c, B_ATTW, nullptr, att, nullptr, nullptr, nullptr, d, nullptr, 0, xab + d, 2 * d, 0, n, d, 0)) != RPG_OK) return rc;
            // node update                                                              my_gnn_layer.py:309-311
            if ((rc = gemm(xab, 2 * d, 2 * d, B_UPD0, tensors[T_UPD0_B], nullptr, nullptr, nullptr, nullptr, 0, nhb, 0, nullptr, 0, 0, n, d, 1)) != RPG_OK) return rc;
            if ((rc = gemm(nhb, d, d, B_UPD2, tensors[T_UPD2_B], nullptr, nullptr, nullptr, nullptr, 0, xnew, 1, xab, 2 * d, 1, n, d, 1)) != RPG_OK) return rc;
            x = xnew;
        }
        if (node_out && hipMemcpyAsync(node_out, x, (size_t)n * d * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            rpg::set_last_error("gnn_forward node_out", hipGetLastError());
            return RPG_ERR_LAUNCH;
        }
        if (edge_out && hipMemcpyAsync(edge_out, ecur, (size_t)e * d * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            rpg::set_last_error("gnn_forward edge_out", hipGetLastError());
            return RPG_ERR_LAUNCH;
        }
        if ((rc = rpg_pose_heads_f32(x, tensors[T_HEADN_W], tensors[T_HEADN_B], n, d, abs_pose, stream)) != RPG_OK) return rc;
        return rpg_pose_heads_f32(ecur, tensors[T_HEADE_W], tensors[T_HEADE_B], e, d, rel_pose, stream);
    }

    // edge_feat = relu(proj_edge(cat[x[min], x[max]]))                                   posenet.py:1053-1055
    const float* x = feat;
    float* ecur = ebuf[0];
    if (split) {
        if ((rc = node_gemm(x, T_PROJN_W, 2 * d)) != RPG_OK) return rc;                  // [n][2d] = [W_lo x | W_hi x]
        if ((rc = rpg::launch_gather_add2_relu(node3, lo, hi, tensors[T_PROJ_B], ecur, e, d, s)) != RPG_OK) return rc;
    } else if ((rc = linear(2, x, lo, d, x, hi, d, nullptr, nullptr, 0, T_PROJ_W, nullptr, ecur, e, d, 1)) != RPG_OK) {
        return rc;
    }

    const bool dual = !wb;                 // fp32: the edge update is stored twice, raw (for the message) and rectified
    for (int r = 0; r < gnn_recursion; ++r) {                                           // posenet.py:1061-1069
        float* enext = (ecur == ebuf[0]) ? ebuf[1] : ebuf[0];    // relu(edge update): the next recursion's / the heads' input
        float* enew = dual ? eraw : enext;                        // the raw edge update (consumed by the message MLP)
        float* xnew = xbuf[r & 1];
        // edge update: edge_mlp(cat[x[src], x[dst], e])                                 my_gnn_layer.py:296-297
        if (split) {
            if ((rc = node_gemm(x, T_NODE3_W, 3 * d)) != RPG_OK) return rc;              // [n][3d] = [Ws x | Wd x | Wm x]
            if ((rc = edge_gemm(ecur, T_EDGE0E_W, T_EDGE0_B, node3, src, node3 + d, dst, hid)) != RPG_OK) return rc;
        } else if ((rc = linear(3, x, src, d, x, dst, d, ecur, nullptr, d, T_EDGE0_W, nullptr, hid, e, d, 1)) != RPG_OK) {
            return rc;
        }
        // edge_feat = relu(edge_feat) of posenet.py:1065 is the second output of this Linear's epilogue (fp32 path)
        if ((rc = linear(1, hid, nullptr, d, nullptr, nullptr, 0, nullptr, nullptr, 0, T_EDGE2_W, nullptr, enew, e, d, 0,
                         dual ? enext : nullptr)) != RPG_OK)
            return rc;
        // message: mlp(cat[x[src], e_new]) then AttentionBlock                          my_gnn_layer.py:304-307
        if (split) {
            if ((rc = edge_gemm(enew, T_MSG0E_W, T_MSG0_B, node3 + 2 * d, src, nullptr, nullptr, hid)) != RPG_OK) return rc;
        } else if ((rc = linear(2, x, src, d, enew, nullptr, d, nullptr, nullptr, 0, T_MSG0_W, nullptr, hid, e, d, 1)) != RPG_OK) {
            return rc;
        }
        if ((rc = linear(1, hid, nullptr, d, nullptr, nullptr, 0, nullptr, nullptr, 0, T_MSG2_W, nullptr, msg, e, d, 0)) != RPG_OK) return rc;
        if ((rc = linear(1, msg, nullptr, d, nullptr, nullptr, 0, nullptr, nullptr, 0, T_GTP_W, nullptr, gtp, e, 3 * c, 0)) != RPG_OK) return rc;
        if (fuse_agg) {
            // aggregate FIRST (att = W y + b + msg is linear in (y, msg): mean(att) = W mean(y) + b + mean(msg)), in the
            // attention kernel itself, then att.W on the n node rows                    my_gnn_layer.py:301,304-307; att.py:32-33
            if ((rc = rpg_attention_aggregate_f32(gtp, msg, rowptr, perm, tensors[T_ATTW_B], n, e, c, d, yat, att, stream)) != RPG_OK)
                return rc;
            if ((rc = linear(1, yat, nullptr, c, nullptr, nullptr, 0, nullptr, nullptr, 0, T_ATTW_W, att, agg, n, d, 0, nullptr,
                             false)) != RPG_OK)
                return rc;
        } else {
            if ((rc = rpg_attention_rows_f32(gtp, e, c, yat, stream)) != RPG_OK) return rc;
            if ((rc = linear(1, yat, nullptr, c, nullptr, nullptr, 0, nullptr, nullptr, 0, T_ATTW_W, msg, att, e, d, 0)) != RPG_OK) return rc;
            // aggregate (mean over incoming edges)                                       my_gnn_layer.py:301
            if ((rc = rpg_scatter_mean_f32(att, rowptr, perm, n, e, d, agg, stream)) != RPG_OK) return rc;
        }
        // node update                                                                   my_gnn_layer.py:309-311
        if ((rc = linear(2, x, nullptr, d, agg, nullptr, d, nullptr, nullptr, 0, T_UPD0_W, nullptr, nhid, n, d, 1)) != RPG_OK) return rc;
        if ((rc = linear(1, nhid, nullptr, d, nullptr, nullptr, 0, nullptr, nullptr, 0, T_UPD2_W, nullptr, xnew, n, d, 1)) != RPG_OK) return rc;
        // x = relu(x) is fused above; edge_feat = relu(edge_feat): dual-stored above, or in place now that the message has
        // consumed the raw one (bf16 Linears)
        if (!dual && (rc = rpg::launch_relu_inplace(enext, (long)e * d, s)) != RPG_OK) return rc;
        x = xnew;
        ecur = enext;
    }
    if (node_out && hipMemcpyAsync(node_out, x, (size_t)n * d * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        rpg::set_last_error("gnn_forward node_out", hipGetLastError());
        return RPG_ERR_LAUNCH;
    }
    if (edge_out && hipMemcpyAsync(edge_out, ecur, (size_t)e * d * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        rpg::set_last_error("gnn_forward edge_out", hipGetLastError());
        return RPG_ERR_LAUNCH;
    }
    // heads (droprate == 0, use_AP)                                                     posenet.py:1077-1091
    if ((rc = rpg_pose_heads_f32(x, tensors[T_HEADN_W], tensors[T_HEADN_B], n, d, abs_pose, stream)) != RPG_OK) return rc;
    return rpg_pose_heads_f32(ecur, tensors[T_HEADE_W], tensors[T_HEADE_B], e, d, rel_pose, stream);
}
}  // namespace

extern "C" int rpg_gnn_forward_f32(const float* const* tensors, int n_tensors, const float* feat, const int64_t* esrc,
                                   const int64_t* edst, int64_t node_offset, int n, int e, int d, int gnn_recursion, float* abs_pose,
                                   float* rel_pose, float* node_out, float* edge_out, int32_t* status, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return gnn_forward_impl(tensors, n_tensors, nullptr, feat, esrc, edst, node_offset, n, e, d, gnn_recursion, abs_pose, rel_pose,
                            node_out, edge_out, status, workspace, workspace_bytes, stream);
}

extern "C" int rpg_gnn_forward_bf16(const float* const* tensors, int n_tensors, const void* const* weights_bf16, int n_bf16,
                                    const float* feat, const int64_t* esrc, const int64_t* edst, int64_t node_offset, int n, int e,
                                    int d, int gnn_recursion, float* abs_pose, float* rel_pose, float* node_out, float* edge_out,
                                    int32_t* status, void* workspace, size_t workspace_bytes, void* stream) {
    if (!weights_bf16 || n_bf16 != B_COUNT || (d & 63)) return RPG_ERR_BAD_ARG;
    for (int i = 0; i < B_COUNT; ++i)
        if (!weights_bf16[i]) return RPG_ERR_BAD_ARG;
    return gnn_forward_impl(tensors, n_tensors, weights_bf16, feat, esrc, edst, node_offset, n, e, d, gnn_recursion, abs_pose,
                            rel_pose, node_out, edge_out, status, workspace, workspace_bytes, stream);
}
