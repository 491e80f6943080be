// vq_models.hpp -- host-side mirror of the reference's model types for the nearest-prototype path.
//
// The reference (GerritLuimstra/vq) is a Rust crate; this image has no cargo/rustc, so the host
// side above the C ABI (include/vqhip.h) is written in C++, header-only, with the reference's own
// names, argument meaning and error behaviour for everything that sits on the path:
//
//     Prototype                                       src/lib.rs:30-34
//     VQ / LVQ / GLVQ ::predict, ::prototypes         src/vq/vq.rs:142-160,183  src/lvq/lvq.rs:193-216
//                                                     src/glvq/glvq.rs:217-240
//     VQ::name_prototypes                             src/vq/vq.rs:170-180
//     GMLVQ ::predict ::prototypes ::omega ::lambda ::project          src/gmlvq/gmlvq.rs:270-372
//     LGMLVQ ::predict ::prototypes ::omegas ::lambdas ::project
//            ::find_closest_local_prototype{,_matched}                 src/lgmlvq/lgmlvq.rs:278-393, 436-506
//     LiRaMLVQ ::predict ::prototypes ::prototypes_omega ::omega ::lambda ::project ::project_omega
//                                                                      src/liramlvq/liramlvq.rs:287-446
//     check_predict_constraints (three asserts, messages pinned by the reference's tests)
//                                                     e.g. src/vq/vq.rs:66-71, tests/glvq.rs:150,220,246
//
// A Rust `panic!` / failed `assert!` becomes a thrown vq::Panic whose what() is the reference's
// message, so tests written against this header read like tests/*.rs (`#[should_panic(expected)]`).
//
// What is NOT here: `fit` (sequential per-sample SGD driven by rand_chacha; SURVEY.md section 8 marks
// it out of scope).  A fitted state is installed with `from_fitted(...)`: the prototypes (and Omega /
// Omegas) a `fit` of the reference would have left in the struct's private fields.  `predict` then is
// the reference's signature with its body replaced: validate -> one batched device call -> index ->
// name.  No batched call in this header computes a distance on the CPU; without libvqhip.so and a gfx950
// device every one of them throws.  (LGMLVQ's two single-sample `pub fn(&self)` searches stay host loops, as in the
// reference: one sample is not a batch.)  `evaluate` is row f3 of SURVEY.md section 8: the fit loops' matched search
// for a whole data set at once, prototypes held fixed.
#ifndef VQ_MODELS_HPP
#define VQ_MODELS_HPP

#include <algorithm>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "vqhip.h"

namespace vq {

using Vector = std::vector<double>;                    // Array1<f64>
using Data = std::vector<Vector>;                      // Vec<Array1<f64>>

struct Matrix {                                        // Array2<f64>, row-major
    size_t rows = 0, cols = 0;
    std::vector<double> a;
    Matrix() = default;
    Matrix(size_t r, size_t c) : rows(r), cols(c), a(r * c, 0.0) {}
    double &operator()(size_t r, size_t c) { return a[r * cols + c]; }
    double operator()(size_t r, size_t c) const { return a[r * cols + c]; }
    Vector dot(const Vector &v) const                  // Array2::dot(&Array1)
    {
        Vector out(rows, 0.0);
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) out[r] += a[r * cols + c] * v[c];
        return out;
    }
    Matrix gram() const                                // omega.t().dot(&omega)  (Lambda)
    {
        Matrix l(cols, cols);
        for (size_t i = 0; i < cols; ++i)
            for (size_t j = 0; j < cols; ++j) {
                double t = 0.0;
                for (size_t r = 0; r < rows; ++r) t += a[r * cols + i] * a[r * cols + j];
                l(i, j) = t;
            }
        return l;
    }
};

struct Prototype {                                     // src/lib.rs:30-34
    Vector vector;
    std::string name;
};

struct Panic : std::runtime_error {                    // a Rust panic: what() is the reference's message
    using std::runtime_error::runtime_error;
};

namespace detail {

inline void rust_assert(bool ok, const char *msg)
{
    if (!ok) throw Panic(msg);
}

// One GPU context per model, created on first use (the reference has no device; a Rust wrapper would
// hold this in the struct and free it in Drop).
class Device {
public:
    explicit Device(int device = 0) : device_(device) {}
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    ~Device()
    {
        if (ctx_) vqhip_destroy(ctx_);
    }
    vqhip_ctx *get()
    {
        if (!ctx_) {
            const int rc = vqhip_create(device_, &ctx_);
            if (rc != VQHIP_OK) throw Panic(std::string("vqhip_create: ") + vqhip_last_error(nullptr));
        }
        return ctx_;
    }
    void check(int rc)
    {
        if (rc != VQHIP_OK) throw Panic(std::string("vqhip: ") + vqhip_last_error(ctx_));
    }

private:
    int device_;
    vqhip_ctx *ctx_ = nullptr;
};

enum class Metric { Euclidean, GlobalOmega, LocalOmegas };

// State and plumbing shared by the six models.
class Base {
protected:
    std::vector<Prototype> prototypes_;                // src/lib.rs: `prototypes: Vec<Prototype>`
    Metric metric_ = Metric::Euclidean;
    Matrix omega_;                                     // GMLVQ / LiRaMLVQ `omega`
    std::vector<Matrix> omegas_;                       // LGMLVQ `omegas`
    std::shared_ptr<Device> dev_ = std::make_shared<Device>(0);
    bool uploaded_ = false;                            // device copy matches prototypes_ / omega(s)

    size_t dim() const { return prototypes_.empty() ? 0 : prototypes_[0].vector.size(); }

    // e.g. src/vq/vq.rs:66-71 -- the three asserts, in the reference's order, before any compute
    void check_predict_constraints(const Data &data) const
    {
        rust_assert(data.size() > 0, "There are no data samples given.");
        rust_assert(prototypes_.size() > 0, "The model has not been fit yet.");
        rust_assert(prototypes_[0].vector.size() == data[0].size(), "Data must have the same dimensions as was used in fit!");
    }

    // Class ids: names interned in sorted order (the BTreeMap order of src/lvq/lvq.rs:101); equal ids
    // <=> equal names, which is all src/helpers.rs:113 compares.
    std::map<std::string, uint32_t> name_table() const
    {
        std::map<std::string, uint32_t> t;
        for (const Prototype &p : prototypes_) t.emplace(p.name, 0u);
        uint32_t next = 0;
        for (auto &kv : t) kv.second = next++;
        return t;
    }

    void upload()
    {
        if (uploaded_) return;
        vqhip_ctx *c = dev_->get();
        const size_t K = prototypes_.size(), D = dim();
        std::vector<float> W(K * D);
        for (size_t k = 0; k < K; ++k) {
            rust_assert(prototypes_[k].vector.size() == D, "assertion failed: vec1.len() == vec2.len()");
            for (size_t d = 0; d < D; ++d) W[k * D + d] = (float)prototypes_[k].vector[d];
        }
        dev_->check(vqhip_set_prototypes(c, W.data(), (uint32_t)K, (uint32_t)D));
        const auto table = name_table();
        std::vector<uint32_t> cls(K);
        for (size_t k = 0; k < K; ++k) cls[k] = table.at(prototypes_[k].name);
        dev_->check(vqhip_set_labels(c, cls.data(), (uint32_t)K));
        if (metric_ == Metric::GlobalOmega) {
            std::vector<float> om(omega_.a.begin(), omega_.a.end());
            dev_->check(vqhip_set_omega(c, om.data(), (uint32_t)omega_.rows, (uint32_t)omega_.cols));
        } else if (metric_ == Metric::LocalOmegas) {
            const size_t M = omegas_[0].rows;
            std::vector<float> om;
            om.reserve(K * M * D);
            for (const Matrix &o : omegas_) om.insert(om.end(), o.a.begin(), o.a.end());
            dev_->check(vqhip_set_local_omegas(c, om.data(), (uint32_t)K, (uint32_t)M, (uint32_t)D));
        }
        uploaded_ = true;
    }

    // Row pointers of a Vec<Array1<f64>>; every row's length is asserted (the reference checks only
    // data[0] up front and panics inside euclidean_distance on a later ragged row, src/helpers.rs:14).
    std::vector<const double *> row_pointers(const Data &data) const
    {
        std::vector<const double *> rows(data.size());
        for (size_t i = 0; i < data.size(); ++i) {
            rust_assert(data[i].size() == dim(), "assertion failed: vec1.len() == vec2.len()");
            rows[i] = data[i].data();
        }
        return rows;
    }

    // find_closest_prototype for every sample in one device call (the loop of src/glvq/glvq.rs:224-232)
    std::vector<uint32_t> closest(const Data &data)
    {
        upload();
        const std::vector<const double *> rows = row_pointers(data);
        std::vector<uint32_t> idx(data.size());
        dev_->check(vqhip_assign_rows_f64(dev_->get(), rows.data(), rows.size(), (uint32_t)dim(), idx.data(), nullptr));
        return idx;
    }

    std::vector<std::string> predict_names(const Data &data)
    {
        check_predict_constraints(data);
        const std::vector<uint32_t> idx = closest(data);
        std::vector<std::string> labels;
        labels.reserve(idx.size());
        for (uint32_t i : idx) labels.push_back(prototypes_[i].name);      // the CURRENT name (tests/vq.rs:229,240)
        return labels;
    }

public:
    // ---- the fit loops' matched search for a whole data set at once ----------------------
    // For every sample: the nearest prototype of its own class (j_plus, d_plus) and the nearest of any other class
    // (j_minus, d_minus) -- the two find_closest_prototype_matched calls and the two re-measured distances of
    // src/glvq/glvq.rs:171-184 (gmlvq.rs / liramlvq.rs / lgmlvq.rs:194-199 with their Omega distances) -- plus what the
    // reference derives from them per sample: the GLVQ cost term mu = (d+ - d-) / (d+ + d-) and the scalar factors of
    // its two prototype updates, 4 d- / (d+ + d-)^2 and 4 d+ / (d+ + d-)^2 (glvq.rs:192-194).
    // This is an EVALUATION with the prototypes held fixed (e.g. at the start of an epoch: cost, training error, the
    // would-be update directions).  It is not `fit`: the reference updates its prototypes after every single sample
    // (glvq.rs:204-205), which a batched call cannot reproduce and does not claim to.
    // One departure, unreachable after check_fit_constraints (every label has a prototype, glvq.rs:63-92): for a sample whose
    // class has NO prototype (or whose class owns all of them) the reference's matched search returns index 0
    // (helpers.rs:98) and fit goes on to measure prototype 0 (glvq.rs:177-184); here such a sample reports index 0 with
    // distance +INF and is left out of cost, error count and gradient.
    struct Evaluation {
        std::vector<uint32_t> j_plus, j_minus;
        std::vector<double> d_plus, d_minus, mu, g_plus, g_minus;
        double cost = 0.0;                             // sum of mu over the samples that have both prototypes
        size_t misclassified = 0;                      // d- < d+  (mu > 0)
    };
    Evaluation evaluate(const Data &data, const std::vector<std::string> &labels)
    {
        check_predict_constraints(data);
        rust_assert(data.size() == labels.size(), "The data vector does not match the label vector in length.");   // glvq.rs:70
        upload();
        const auto table = name_table();
        std::vector<uint32_t> y(labels.size());
        for (size_t i = 0; i < labels.size(); ++i) {
            const auto it = table.find(labels[i]);
            y[i] = it == table.end() ? 0xFFFFFFFFu : it->second;               // an unknown label matches no prototype
        }
        const std::vector<const double *> rows = row_pointers(data);
        const size_t N = data.size();
        Evaluation e;
        e.j_plus.resize(N); e.j_minus.resize(N);
        std::vector<float> dp(N), dn(N);
        dev_->check(vqhip_assign_matched_rows_f64(dev_->get(), rows.data(), y.data(), N, (uint32_t)dim(), e.j_plus.data(), dp.data(),
                                                  e.j_minus.data(), dn.data()));
        e.d_plus.assign(dp.begin(), dp.end());
        e.d_minus.assign(dn.begin(), dn.end());
        e.mu.assign(N, 0.0); e.g_plus.assign(N, 0.0); e.g_minus.assign(N, 0.0);
        for (size_t i = 0; i < N; ++i) {
            const double d1 = e.d_plus[i], d2 = e.d_minus[i], s = d1 + d2;
            if (!(s > 0.0) || !(s < 1e300)) continue;                         // no prototype of / outside the class, or both at distance 0
            e.mu[i] = (d1 - d2) / s;
            e.g_plus[i] = 4.0 * d2 / (s * s);
            e.g_minus[i] = 4.0 * d1 / (s * s);
            e.cost += e.mu[i];
            e.misclassified += d2 < d1;
        }
        return e;
    }

    // The batch form of the two GLVQ update terms (vqhip_glvq_batch_gradient, include/vqhip.h): dW[k] = sum over the samples
    // of glvq.rs:193-194 for the prototypes they would move, with the prototypes held fixed.  NOT the reference's `fit`
    // (per-sample updates, glvq.rs:204-205): a full-batch step W <- W - lr * dW is a different algorithm.  Euclidean models.
    struct BatchGradient {
        std::vector<Vector> dW;                        // K rows of D values
        double cost = 0.0;
        size_t used = 0;                               // samples that have both prototypes and d+ + d- > 0
    };
    BatchGradient batch_gradient(const Data &data, const std::vector<std::string> &labels)
    {
        check_predict_constraints(data);
        rust_assert(data.size() == labels.size(), "The data vector does not match the label vector in length.");   // glvq.rs:70
        upload();
        const auto table = name_table();
        const size_t N = data.size(), D = dim(), K = prototypes_.size();
        std::vector<uint32_t> y(N);
        std::vector<float> X(N * D), G(K * D);
        for (size_t i = 0; i < N; ++i) {
            rust_assert(data[i].size() == D, "assertion failed: vec1.len() == vec2.len()");
            for (size_t d = 0; d < D; ++d) X[i * D + d] = (float)data[i][d];
            const auto it = table.find(labels[i]);
            y[i] = it == table.end() ? 0xFFFFFFFFu : it->second;
        }
        BatchGradient g;
        uint64_t used = 0;
        dev_->check(vqhip_glvq_batch_gradient(dev_->get(), X.data(), y.data(), N, G.data(), &g.cost, &used));
        g.used = (size_t)used;
        g.dW.assign(K, Vector(D));
        for (size_t k = 0; k < K; ++k)
            for (size_t d = 0; d < D; ++d) g.dW[k][d] = G[k * D + d];
        return g;
    }

protected:
    Data project_with(const Data &data, int which)
    {
        upload();
        const size_t N = data.size(), D = dim();
        const size_t width = which == VQHIP_PROJECT_OMEGA ? omega_.rows : D;
        // the reference's container on both sides: the library gathers / narrows and widens / scatters in its pinned
        // buffers (vqhip_project_rows_f64), no contiguous copy of either matrix here
        Data out(N, Vector(width));
        std::vector<const double *> rows(N);
        std::vector<double *> out_rows(N);
        for (size_t i = 0; i < N; ++i) {
            rust_assert(data[i].size() == D, "assertion failed: vec1.len() == vec2.len()");
            rows[i] = data[i].data();
            out_rows[i] = out[i].data();
        }
        dev_->check(vqhip_project_rows_f64(dev_->get(), rows.data(), N, (uint32_t)D, which, out_rows.data()));
        return out;
    }

    std::vector<Prototype> projected_prototypes(const char *msg) const         // e.g. src/gmlvq/gmlvq.rs:298-323
    {
        rust_assert(prototypes_.size() > 0, msg);
        std::vector<Prototype> out;
        const Matrix lambda = metric_ == Metric::GlobalOmega ? omega_.gram() : Matrix();
        for (size_t k = 0; k < prototypes_.size(); ++k) {
            Prototype p = prototypes_[k];
            p.vector = (metric_ == Metric::GlobalOmega ? lambda : omegas_[k].gram()).dot(p.vector);
            out.push_back(std::move(p));
        }
        return out;
    }

    void install(std::vector<Prototype> prototypes)
    {
        rust_assert(prototypes_.size() == 0, "This model has already been fit.");   // src/vq/vq.rs:59
        rust_assert(prototypes.size() > 0, "There are no data samples given.");
        prototypes_ = std::move(prototypes);
        uploaded_ = false;
    }

public:
    // Which GPU serves this model (default 0).  Replaces nothing in the reference.
    void set_device(int device)
    {
        dev_ = std::make_shared<Device>(device);
        uploaded_ = false;
    }
};

#define VQ_NOT_FIT(what) "The model has not been fit yet. \n\n        " what

}  // namespace detail

// ------------------------------------------------------------------------------------------- Euclidean models
class VQ : public detail::Base {                       // src/lib.rs:47, src/vq/vq.rs
public:
    // The state `fit` leaves behind (src/vq/vq.rs:76-90 names the prototypes "0", "1", ...).
    static VQ from_fitted(std::vector<Prototype> prototypes)
    {
        VQ m;
        m.install(std::move(prototypes));
        return m;
    }
    std::vector<std::string> predict(const Data &data) { return predict_names(data); }   // src/vq/vq.rs:142-160
    void name_prototypes(const std::vector<std::string> &names)                        // src/vq/vq.rs:170-180
    {
        detail::rust_assert(prototypes_.size() > 0, "The model has not been fit yet.");
        detail::rust_assert(names.size() == prototypes_.size(),
                            "The size of the names vectors does not match the amount of the prototypes.");
        for (size_t i = 0; i < names.size(); ++i) prototypes_[i].name = names[i];
        uploaded_ = false;                             // class ids follow the names; the vectors do not change
    }
    const std::vector<Prototype> &prototypes() const { return prototypes_; }           // src/vq/vq.rs:183
};

class LVQ : public detail::Base {                      // src/lib.rs:73, src/lvq/lvq.rs
public:
    static LVQ from_fitted(std::vector<Prototype> prototypes)
    {
        LVQ m;
        m.install(std::move(prototypes));
        return m;
    }
    std::vector<std::string> predict(const Data &data) { return predict_names(data); }   // src/lvq/lvq.rs:193-211
    const std::vector<Prototype> &prototypes() const { return prototypes_; }           // src/lvq/lvq.rs:214
};

class GLVQ : public detail::Base {                     // src/lib.rs:108, src/glvq/glvq.rs
public:
    static GLVQ from_fitted(std::vector<Prototype> prototypes)
    {
        GLVQ m;
        m.install(std::move(prototypes));
        return m;
    }
    std::vector<std::string> predict(const Data &data) { return predict_names(data); }   // src/glvq/glvq.rs:217-235
    const std::vector<Prototype> &prototypes() const { return prototypes_; }           // src/glvq/glvq.rs:238
};

// ------------------------------------------------------------------------------------------- global Omega
class GMLVQ : public detail::Base {                    // src/lib.rs:183, src/gmlvq/gmlvq.rs
public:
    static GMLVQ from_fitted(std::vector<Prototype> prototypes, Matrix omega)
    {
        GMLVQ m;
        m.install(std::move(prototypes));
        detail::rust_assert(omega.cols == m.dim() && omega.rows > 0, "Omega must be M x D.");
        m.metric_ = detail::Metric::GlobalOmega;
        m.omega_ = std::move(omega);
        return m;
    }
    std::vector<std::string> predict(const Data &data) { return predict_names(data); }   // src/gmlvq/gmlvq.rs:270-290
    std::vector<Prototype> prototypes() const                                          // :298-323 (Lambda . w)
    {
        return projected_prototypes(VQ_NOT_FIT("There are no prototypes at this stage."));
    }
    const Matrix &omega() const                                                         // :326-333
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("Omega is not available yet at this stage."));
        return omega_;
    }
    Matrix lambda() const                                                               // :336-346
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("Lambda is not available yet at this stage."));
        return omega_.gram();
    }
    Data project(const Data &data) { return project_with(data, VQHIP_PROJECT_LAMBDA); }  // :354-372
};

class LiRaMLVQ : public detail::Base {                 // src/lib.rs:266, src/liramlvq/liramlvq.rs
public:
    static LiRaMLVQ from_fitted(std::vector<Prototype> prototypes, Matrix omega)        // omega: max_rank x D
    {
        LiRaMLVQ m;
        m.install(std::move(prototypes));
        detail::rust_assert(omega.rows > 0, "The max rank needs to be bigger than 0!");  // src/liramlvq/liramlvq.rs:80
        detail::rust_assert(omega.cols == m.dim(), "Omega must be M x D.");
        m.metric_ = detail::Metric::GlobalOmega;
        m.omega_ = std::move(omega);
        return m;
    }
    std::vector<std::string> predict(const Data &data) { return predict_names(data); }   // :287-307
    const Matrix &omega() const                                                         // :310-317
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("Omega is not available yet at this stage."));
        return omega_;
    }
    Matrix lambda() const                                                               // :320-330
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("Lambda is not available yet at this stage."));
        return omega_.gram();
    }
    std::vector<Prototype> prototypes() const                                          // :336-361 (Lambda . w)
    {
        return projected_prototypes(VQ_NOT_FIT("There are no prototypes at this stage."));
    }
    std::vector<Prototype> prototypes_omega() const                                    // :368-392 (Omega . w)
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("There are no prototypes at this stage."));
        std::vector<Prototype> out = prototypes_;
        for (Prototype &p : out) p.vector = omega_.dot(p.vector);
        return out;
    }
    Data project(const Data &data) { return project_with(data, VQHIP_PROJECT_LAMBDA); }        // :401-419
    Data project_omega(const Data &data) { return project_with(data, VQHIP_PROJECT_OMEGA); }   // :429-446
};

// ------------------------------------------------------------------------------------------- local Omegas
class LGMLVQ : public detail::Base {                   // src/lib.rs:223, src/lgmlvq/lgmlvq.rs
public:
    static LGMLVQ from_fitted(std::vector<Prototype> prototypes, std::vector<Matrix> omegas)
    {
        LGMLVQ m;
        m.install(std::move(prototypes));
        detail::rust_assert(omegas.size() == m.prototypes_.size(), "One Omega per prototype (src/lgmlvq/lgmlvq.rs:153-158).");
        for (const Matrix &o : omegas)
            detail::rust_assert(o.cols == m.dim() && o.rows == omegas[0].rows && o.rows > 0, "Every Omega must be M x D.");
        m.metric_ = detail::Metric::LocalOmegas;
        m.omegas_ = std::move(omegas);
        return m;
    }
    std::vector<std::string> predict(const Data &data) { return predict_names(data); }   // :278-298
    std::vector<Prototype> prototypes() const                                          // :306-331 (Lambda_j . w_j)
    {
        return projected_prototypes(VQ_NOT_FIT("There are no prototypes at this stage."));
    }
    const std::vector<Matrix> &omegas() const                                           // :334-341
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("The omegas are not available yet at this stage."));
        return omegas_;
    }
    std::vector<Matrix> lambdas() const                                                 // :344-361
    {
        detail::rust_assert(prototypes_.size() > 0, VQ_NOT_FIT("The Lambdas are not available yet at this stage."));
        std::vector<Matrix> out;
        for (const Matrix &o : omegas_) out.push_back(o.gram());
        return out;
    }
    Data project(const Data &data) { return project_with(data, VQHIP_PROJECT_LAMBDA); }  // :372-393

    // The two `pub` single-sample searches (:436-455, :468-506), `&self` in the reference and const here.  One sample
    // is not a batch: a launch through pinned staging costs more than the K quadratic forms it would replace, so these
    // stay on the host like the reference's helpers stay for `fit` -- the same loop, f64,
    // Lambda_j = Omega_j^T Omega_j rebuilt per prototype as src/helpers.rs:38 does.  Batches go through predict / evaluate.
    size_t find_closest_local_prototype(const Vector &sample) const { return host_search(sample, nullptr, true); }
    size_t find_closest_local_prototype_matched(const Vector &sample, const std::string &label, bool find_closest_matching) const
    {
        return host_search(sample, &label, find_closest_matching);
    }

private:
    size_t host_search(const Vector &sample, const std::string *label, bool matching) const
    {
        detail::rust_assert(sample.size() == dim(), "assertion failed: vec1.len() == vec2.len()");
        size_t best = 0;
        double smallest = std::numeric_limits<double>::infinity();               // lgmlvq.rs:439-440
        for (size_t j = 0; j < prototypes_.size(); ++j) {
            Vector diff(sample.size());
            for (size_t d = 0; d < diff.size(); ++d) diff[d] = sample[d] - prototypes_[j].vector[d];   // helpers.rs:41
            const Vector ld = omegas_[j].gram().dot(diff);                                             // helpers.rs:38, :44
            double dist = 0.0;
            for (size_t d = 0; d < diff.size(); ++d) dist += diff[d] * ld[d];
            const bool eligible = !label || (prototypes_[j].name == *label) == matching;               // lgmlvq.rs:487 / :496
            if (dist < smallest && eligible) {                                                         // strict '<': lgmlvq.rs:448
                smallest = dist;
                best = j;
            }
        }
        return best;
    }

public:
};

#undef VQ_NOT_FIT

}  // namespace vq
#endif  // VQ_MODELS_HPP
