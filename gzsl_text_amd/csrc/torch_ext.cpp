// TORCH_LIBRARY form of the hot path's operators: a thin C++ extension over the C ABI (include/gzsl_hip.h).
//
// SURVEY 8b / north_star name "a thin C++ extension" as the boundary; the ctypes host mirror (gzsl_text_amd/modules) is
// one binding of libgzsl_hip.so, this file is the other.  It registers differentiable torch operators that a maintainer
// of the reference can call from the UNCHANGED modules:
//   gzsl::label_attention  replaces modules/icd_modules.py:137-142 (scores, softmax over tokens, attn dropout, pooling)
//                          and its autograd; the B x L x T' score tensor is never formed
//   gzsl::label_logits     replaces modules/icd_modules.py:177-182 (relu, multiply by the label embedding, sum)
//   gzsl::ldam_bce         replaces modules/icd_modules.py:162-172 + BCEWithLogitsLoss(sum) / B (common_module.py:29-30)
// Tensors are allocated here with ATen (the C ABI itself never allocates); launches go to the CURRENT HIP stream of the
// tensors' device.  Built with g++ (no device code): __graft_entry__.build() -> gzsl_text_amd/lib/libgzsl_torch.so.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/gzsl_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check_f32_cuda(const Tensor& t, const char* name, int64_t dims) {
    TORCH_CHECK(t.is_cuda(), "gzsl: ", name, " must be a GPU tensor (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, "gzsl: ", name, " must be float32");
    TORCH_CHECK(t.dim() == dims, "gzsl: ", name, " must have ", dims, " dimensions");
}
void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, "gzsl: ", what, " failed with code ", rc); }
// (ROCm builds of torch present the GPU as device type "cuda": the stream getter is the masquerading one)
void* cur_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }
const float* cptr(const Tensor& t) { return t.data_ptr<float>(); }

// (threshold, keep scale) of the counter RNG for drop probability p -- ops_hip.drop_params / oracle/rng.py
void drop_params(double p, uint32_t& thresh, float& scale) {
    if (p <= 0.0) {
        thresh = 0, scale = 1.f;
        return;
    }
    const double t = p * 4294967296.0;
    thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    scale = (float)(1.0 / (1.0 - p));
}

// label splits of the backward launch: enough workgroups for 256 CUs (the rule of ops_hip.HipOps.attn_bwd)
int bwd_nsplit(int64_t B, int64_t Tp, int64_t E, int64_t L) {
    const int64_t tok = E <= 100 ? 128 : 64, wgs = B * ((Tp + tok - 1) / tok);
    int ns = 1;
    while (wgs * ns < 1024 && (int64_t)ns * 2 * 16 <= L) ns *= 2;
    return ns;
}

struct LabelAttention : public torch::autograd::Function<LabelAttention> {
    static variable_list forward(AutogradContext* ctx, const Tensor& keys, const Tensor& values, const Tensor& queries,
                                 double p_drop, int64_t seed, int64_t l_offset, int64_t L_total) {
        check_f32_cuda(keys, "keys", 3);
        check_f32_cuda(values, "values", 3);
        check_f32_cuda(queries, "queries", 2);
        TORCH_CHECK(keys.size(0) == values.size(0) && keys.size(1) == values.size(1), "gzsl: keys / values disagree");
        TORCH_CHECK(keys.size(2) == queries.size(1), "gzsl: keys / queries disagree on the embedding width");
        TORCH_CHECK(keys.device() == values.device() && keys.device() == queries.device(), "gzsl: tensors on one device");
        const c10::DeviceGuard guard(keys.device());
        const Tensor PX = keys.contiguous(), X = values.contiguous(), Q = queries.contiguous();
        const int64_t B = PX.size(0), Tp = PX.size(1), E = PX.size(2), F = X.size(2), L = Q.size(0);
        if (L_total <= 0) L_total = L;
        Tensor O = at::empty({B, L, F}, PX.options()), lse = at::empty({B, L}, PX.options());
        uint32_t th;
        float sc;
        drop_params(p_drop, th, sc);
        check_rc(gzsl_label_attn_fwd(cptr(PX), cptr(X), cptr(Q), O.data_ptr<float>(), lse.data_ptr<float>(), (int)B, (int)Tp,
                                     (int)E, (int)F, (int)L, (int)l_offset, (int)L_total, (uint32_t)seed, th, sc, cur_stream()),
                 "gzsl_label_attn_fwd");
        ctx->save_for_backward({PX, X, Q, O, lse});
        ctx->saved_data["p"] = p_drop;
        ctx->saved_data["seed"] = seed;
        ctx->saved_data["l_offset"] = l_offset;
        ctx->saved_data["L_total"] = L_total;
        ctx->mark_non_differentiable({lse});
        return {O, lse};
    }

    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &PX = saved[0], &X = saved[1], &Q = saved[2], &O = saved[3], &lse = saved[4];
        const c10::DeviceGuard guard(PX.device());
        const Tensor dO = grads[0].contiguous();
        const int64_t B = PX.size(0), Tp = PX.size(1), E = PX.size(2), F = X.size(2), L = Q.size(0);
        Tensor delta = at::empty({B, L}, PX.options());
        Tensor dPX = at::zeros_like(PX), dX = at::zeros_like(X), dQ = at::zeros_like(Q);
        uint32_t th;
        float sc;
        drop_params(ctx->saved_data["p"].toDouble(), th, sc);
        void* st = cur_stream();
        check_rc(gzsl_attn_delta(cptr(dO), cptr(O), delta.data_ptr<float>(), B * L, (int)F, st), "gzsl_attn_delta");
        check_rc(gzsl_label_attn_bwd(cptr(dO), cptr(PX), cptr(X), cptr(Q), cptr(lse), cptr(delta), dPX.data_ptr<float>(),
                                     dX.data_ptr<float>(), dQ.data_ptr<float>(), (int)B, (int)Tp, (int)E, (int)F, (int)L,
                                     (int)ctx->saved_data["l_offset"].toInt(), (int)ctx->saved_data["L_total"].toInt(),
                                     bwd_nsplit(B, Tp, E, L), (uint32_t)ctx->saved_data["seed"].toInt(), th, sc, st),
                 "gzsl_label_attn_bwd");
        return {dPX, dX, dQ, Tensor(), Tensor(), Tensor(), Tensor()};
    }
};

std::tuple<Tensor, Tensor> label_attention(const Tensor& keys, const Tensor& values, const Tensor& queries, double p_drop,
                                           int64_t seed, int64_t l_offset, int64_t L_total) {
    auto out = LabelAttention::apply(keys, values, queries, p_drop, seed, l_offset, L_total);
    return {out[0], out[1]};
}

struct LabelLogits : public torch::autograd::Function<LabelLogits> {
    static Tensor forward(AutogradContext* ctx, const Tensor& direct, const Tensor& clf) {
        check_f32_cuda(direct, "direct", 3);
        check_f32_cuda(clf, "label_emb", 2);
        TORCH_CHECK(direct.size(1) == clf.size(0) && direct.size(2) == clf.size(1), "gzsl: direct [B,L,D] vs label_emb [L,D]");
        const c10::DeviceGuard guard(direct.device());
        const Tensor d = direct.contiguous(), c = clf.contiguous();
        Tensor logits = at::empty({d.size(0), d.size(1)}, d.options());
        check_rc(gzsl_logits_fwd(cptr(d), cptr(c), logits.data_ptr<float>(), (int)d.size(0), (int)d.size(1), (int)d.size(2),
                                 cur_stream()),
                 "gzsl_logits_fwd");
        ctx->save_for_backward({d, c});
        return logits;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const Tensor &d = saved[0], &c = saved[1];
        const c10::DeviceGuard guard(d.device());
        const Tensor dl = grads[0].contiguous();
        Tensor dd = at::empty_like(d), dc = at::empty_like(c);
        check_rc(gzsl_logits_bwd(cptr(d), cptr(c), cptr(dl), dd.data_ptr<float>(), dc.data_ptr<float>(), (int)d.size(0),
                                 (int)d.size(1), (int)d.size(2), 0, cur_stream()),
                 "gzsl_logits_bwd");
        return {dd, dc};
    }
};
Tensor label_logits(const Tensor& direct, const Tensor& clf) { return LabelLogits::apply(direct, clf); }

struct LdamBce : public torch::autograd::Function<LdamBce> {
    static Tensor forward(AutogradContext* ctx, const Tensor& logits, const Tensor& y, const c10::optional<Tensor>& margin,
                          double C) {
        check_f32_cuda(logits, "logits", 2);
        check_f32_cuda(y, "y", 2);
        TORCH_CHECK(logits.sizes() == y.sizes(), "gzsl: logits / y shapes differ");
        const c10::DeviceGuard guard(logits.device());
        const Tensor z = logits.contiguous(), yy = y.contiguous();
        Tensor m;
        if (margin.has_value() && margin->defined() && C > 0.0) {
            check_f32_cuda(*margin, "class_margin", 1);
            TORCH_CHECK(margin->size(0) == z.size(1), "gzsl: class_margin has one entry per label");
            m = margin->contiguous();
        }
        Tensor loss = at::zeros({}, z.options()), dlogits = at::empty_like(z);
        check_rc(gzsl_ldam_bce(cptr(z), cptr(yy), m.defined() ? cptr(m) : nullptr, (float)C, 1.f / (float)z.size(0),
                               loss.data_ptr<float>(), dlogits.data_ptr<float>(), nullptr, (int)z.size(0), (int)z.size(1),
                               cur_stream()),
                 "gzsl_ldam_bce");
        ctx->save_for_backward({dlogits});
        return loss;
    }
    static variable_list backward(AutogradContext* ctx, variable_list grads) {
        const Tensor dlogits = ctx->get_saved_variables()[0];
        return {dlogits * grads[0], Tensor(), Tensor(), Tensor()};
    }
};
Tensor ldam_bce(const Tensor& logits, const Tensor& y, const c10::optional<Tensor>& margin, double C) {
    return LdamBce::apply(logits, y, margin, C);
}

}  // namespace

TORCH_LIBRARY(gzsl, m) {
    m.def("label_attention(Tensor keys, Tensor values, Tensor queries, float p_drop=0.0, int seed=0, int l_offset=0, "
          "int L_total=0) -> (Tensor, Tensor)");
    m.def("label_logits(Tensor direct, Tensor label_emb) -> Tensor");
    m.def("ldam_bce(Tensor logits, Tensor y, Tensor? class_margin=None, float C=0.0) -> Tensor");
}

// The autograd::Function::apply wrappers record the graph themselves; they are registered for the Autograd key so that
// tensors that require grad reach them, and for the backend key for calls under no_grad.
TORCH_LIBRARY_IMPL(gzsl, Autograd, m) {
    m.impl("label_attention", label_attention);
    m.impl("label_logits", label_logits);
    m.impl("ldam_bce", ldam_bce);
}
TORCH_LIBRARY_IMPL(gzsl, CUDA, m) {
    m.impl("label_attention", label_attention);
    m.impl("label_logits", label_logits);
    m.impl("ldam_bce", ldam_bce);
}
