// Tests of include/vq_models.hpp, shaped like the reference's tests/*.rs for the predict path:
// the three `check_constraints_predict_*` panics per model (tests/glvq.rs:150,220,246 and twins),
// the toy clustering / classification sets with the prototypes `fit` is asserted to reach
// (tests/vq.rs:164-202, tests/lvq.rs:283-334, tests/glvq.rs:283-325, ...), the renaming test
// (tests/vq.rs:205-241), and the accessors.  `fit` itself is out of scope, so a
// fitted state is installed with from_fitted(...).
//
// Usage: test_models            run everything on GPU 0
//        test_models --list     print the test names (no GPU needed)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "vq_models.hpp"

using namespace vq;

namespace {
int g_failures = 0;
struct Test { const char *name; std::function<void()> fn; };
std::vector<Test> &tests() { static std::vector<Test> t; return t; }
struct Reg { Reg(const char *n, std::function<void()> f) { tests().push_back({n, std::move(f)}); } };
#define TEST(name) void name(); Reg reg_##name(#name, name); void name()
#define CHECK(cond) do { if (!(cond)) { std::printf("    CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); ++g_failures; } } while (0)

// #[should_panic(expected = "...")]: the message must CONTAIN `expected`
template <class F> void should_panic(const char *expected, F &&f)
{
    try {
        f();
    } catch (const Panic &p) {
        if (!std::strstr(p.what(), expected)) { std::printf("    panicked with \"%s\", expected \"%s\"\n", p.what(), expected); ++g_failures; }
        return;
    }
    std::printf("    did not panic (expected \"%s\")\n", expected);
    ++g_failures;
}

bool vec_all_close(const Vector &a, const Vector &b, double tol)          // tests/vq.rs:5-13
{
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (std::fabs(a[i] - b[i]) > tol) return false;
    return true;
}

using Names = std::vector<std::string>;
const Data TOY = {{5.0, 2.0}, {5.0, 3.0}, {5.0, 4.0}, {-5.0, 10.0}, {-5.0, 11.0}, {-5.0, 12.0}};   // tests/vq.rs:168-175
Matrix identity_over_sqrt_n(size_t n)                                      // src/gmlvq/gmlvq.rs:160-161 after normalize_omega
{
    Matrix o(n, n);
    for (size_t i = 0; i < n; ++i) o(i, i) = 1.0 / std::sqrt((double)n);
    return o;
}
std::vector<Prototype> lvq_prototypes() { return {{{5.0, 3.0}, "C0"}, {{-5.0, 11.0}, "C1"}}; }       // tests/lvq.rs:332-333
}  // namespace

// ---------------------------------------------------------------------------------- predict constraints
template <class M> void constraints(M fitted)
{
    should_panic("There are no data samples given.", [&] { fitted.predict(Data{}); });                  // tests/glvq.rs:150
    should_panic("The model has not been fit yet.", [&] { M unfit; unfit.predict(Data{{5.0, 2.0}}); });    // tests/glvq.rs:220
    should_panic("Data must have the same dimensions as was used in fit!", [&] { fitted.predict(Data{{5.0, 2.0, 1.0}}); });   // :246
    // a ragged row after the first: the reference panics inside euclidean_distance (src/helpers.rs:14)
    should_panic("vec1.len() == vec2.len()", [&] { fitted.predict(Data{{5.0, 2.0}, {5.0}}); });
    // no data is checked before "not fit" (order of the asserts, src/vq/vq.rs:67-68)
    should_panic("There are no data samples given.", [&] { M unfit; unfit.predict(Data{}); });
}
TEST(check_constraints_predict_vq) { constraints(VQ::from_fitted({{{-5.0, 11.0}, "0"}, {{5.0, 3.0}, "1"}})); }
TEST(check_constraints_predict_lvq) { constraints(LVQ::from_fitted(lvq_prototypes())); }
TEST(check_constraints_predict_glvq) { constraints(GLVQ::from_fitted(lvq_prototypes())); }
TEST(check_constraints_predict_gmlvq) { constraints(GMLVQ::from_fitted(lvq_prototypes(), identity_over_sqrt_n(2))); }
TEST(check_constraints_predict_lgmlvq)
{
    constraints(LGMLVQ::from_fitted(lvq_prototypes(), {identity_over_sqrt_n(2), identity_over_sqrt_n(2)}));
}
TEST(check_constraints_predict_liramlvq)
{
    Matrix om(1, 2);
    om(0, 0) = 0.6; om(0, 1) = -0.8;
    constraints(LiRaMLVQ::from_fitted(lvq_prototypes(), om));
}
TEST(check_constraints_max_rank)                                           // tests/liramlvq.rs:17
{
    should_panic("The max rank needs to be bigger than 0!", [] { LiRaMLVQ::from_fitted(lvq_prototypes(), Matrix(0, 2)); });   // tests/liramlvq.rs:17
}

// ---------------------------------------------------------------------------------- end to end on the toy sets
TEST(simple_clustering_vq)                                                 // tests/vq.rs:164-202
{
    VQ model = VQ::from_fitted({{{-5.0, 11.0}, "0"}, {{5.0, 3.0}, "1"}});   // the centres asserted at :200-201
    CHECK(model.predict(TOY) == (Names{"1", "1", "1", "0", "0", "0"}));     // :192
    const auto &p = model.prototypes();
    CHECK(vec_all_close(p[0].vector, {-5.0, 11.0}, 1e-1));
    CHECK(vec_all_close(p[1].vector, {5.0, 3.0}, 1e-1));
}
TEST(renaming_prototypes)                                                  // tests/vq.rs:205-241
{
    const Data data = {{5.0, 2.0}, {5.0, 3.0}};
    VQ model = VQ::from_fitted({{{5.0, 2.0}, "0"}, {{5.0, 3.0}, "1"}});
    CHECK(model.predict(data) == (Names{"0", "1"}));                        // :229
    model.name_prototypes({"class 0", "class 1"});
    CHECK(model.predict(data) == (Names{"class 0", "class 1"}));            // :240
    should_panic("The size of the names vectors does not match the amount of the prototypes.", [&] { model.name_prototypes({"x"}); });
    should_panic("The model has not been fit yet.", [] { VQ unfit; unfit.name_prototypes({"a"}); });
}
TEST(simple_classification_lvq)                                            // tests/lvq.rs:283-334
{
    LVQ model = LVQ::from_fitted(lvq_prototypes());
    CHECK(model.predict(TOY) == (Names{"C0", "C0", "C0", "C1", "C1", "C1"}));   // :324
    CHECK(vec_all_close(model.prototypes()[0].vector, {5.0, 3.0}, 1e-1));
    CHECK(vec_all_close(model.prototypes()[1].vector, {-5.0, 11.0}, 1e-1));
}
TEST(simple_classification_glvq)                                           // tests/glvq.rs:283-325
{
    GLVQ model = GLVQ::from_fitted(lvq_prototypes());
    CHECK(model.predict(TOY) == (Names{"C0", "C0", "C0", "C1", "C1", "C1"}));
}
TEST(simple_classification_gmlvq)                                          // tests/gmlvq.rs:283-325
{
    GMLVQ model = GMLVQ::from_fitted(lvq_prototypes(), identity_over_sqrt_n(2));
    CHECK(model.predict(TOY) == (Names{"C0", "C0", "C0", "C1", "C1", "C1"}));
    // Lambda = I / 2: prototypes() returns Lambda . w (src/gmlvq/gmlvq.rs:298-323), project the same map
    CHECK(vec_all_close(model.prototypes()[0].vector, {2.5, 1.5}, 1e-12));
    CHECK(vec_all_close(model.prototypes()[1].vector, {-2.5, 5.5}, 1e-12));
    const Matrix l = model.lambda();
    CHECK(std::fabs(l(0, 0) - 0.5) < 1e-12 && std::fabs(l(1, 1) - 0.5) < 1e-12 && l(0, 1) == 0.0);
    CHECK(model.omega().rows == 2);
    const Data y = model.project(TOY);
    CHECK(y.size() == 6 && vec_all_close(y[0], {2.5, 1.0}, 1e-5) && vec_all_close(y[5], {-2.5, 6.0}, 1e-5));
    should_panic("The model has not been fit yet.", [] { GMLVQ unfit; unfit.prototypes(); });
    should_panic("Omega is not available yet at this stage.", [] { GMLVQ unfit; unfit.omega(); });
    should_panic("Lambda is not available yet at this stage.", [] { GMLVQ unfit; unfit.lambda(); });
}
TEST(simple_classification_lgmlvq)                                         // tests/lgmlvq.rs:283-325
{
    LGMLVQ model = LGMLVQ::from_fitted(lvq_prototypes(), {identity_over_sqrt_n(2), identity_over_sqrt_n(2)});
    CHECK(model.predict(TOY) == (Names{"C0", "C0", "C0", "C1", "C1", "C1"}));
    CHECK(model.find_closest_local_prototype({5.0, 2.0}) == 0);             // src/lgmlvq/lgmlvq.rs:436-455
    CHECK(model.find_closest_local_prototype({-5.0, 12.0}) == 1);
    // the matched twins (:468-506): nearest prototype WITH / WITHOUT the label, 0 when none is eligible
    CHECK(model.find_closest_local_prototype_matched({5.0, 2.0}, "C1", true) == 1);
    CHECK(model.find_closest_local_prototype_matched({5.0, 2.0}, "C1", false) == 0);
    CHECK(model.find_closest_local_prototype_matched({5.0, 2.0}, "nope", true) == 0);
    CHECK(model.omegas().size() == 2 && model.lambdas().size() == 2);
    CHECK(vec_all_close(model.prototypes()[1].vector, {-2.5, 5.5}, 1e-12));
    const Data y = model.project(TOY);
    CHECK(vec_all_close(y[1], {2.5, 1.5}, 1e-5));
    should_panic("The omegas are not available yet at this stage.", [] { LGMLVQ unfit; unfit.omegas(); });
}
TEST(lgmlvq_local_omega_is_honoured)                                       // per-prototype Omega decides
{
    Matrix big(2, 2);
    big(0, 0) = big(1, 1) = 10.0;
    LGMLVQ model = LGMLVQ::from_fitted({{{0.0, 0.0}, "far"}, {{3.0, 0.0}, "near"}}, {identity_over_sqrt_n(2), big});
    // x = (2, 0): Euclid picks "near" (distance 1 vs 2); with Omega_1 = 10 I its distance is 100 against 2
    CHECK(model.predict(Data{{2.0, 0.0}}) == (Names{"far"}));
}
TEST(simple_classification_liramlvq)                                       // tests/liramlvq.rs:325-368
{
    Matrix om(1, 2);                                                        // rank 1, unit Frobenius norm
    om(0, 0) = 0.6; om(0, 1) = -0.8;
    LiRaMLVQ model = LiRaMLVQ::from_fitted(lvq_prototypes(), om);
    CHECK(model.predict(TOY) == (Names{"C0", "C0", "C0", "C1", "C1", "C1"}));   // :367
    const Data z = model.project_omega(TOY);                                // Omega x: 1-d
    CHECK(z[0].size() == 1 && std::fabs(z[0][0] - (3.0 - 1.6)) < 1e-5);
    const Data y = model.project(TOY);                                      // Lambda x = Omega^T (Omega x)
    CHECK(vec_all_close(y[0], {0.6 * 1.4, -0.8 * 1.4}, 1e-5));
    CHECK(vec_all_close(model.prototypes_omega()[0].vector, {0.6 * 5.0 - 0.8 * 3.0}, 1e-12));
    CHECK(vec_all_close(model.prototypes()[0].vector, {0.6 * 0.6, -0.8 * 0.6}, 1e-12));
    CHECK(std::fabs(model.lambda()(0, 1) + 0.48) < 1e-12);
}
TEST(glvq_evaluate_matched_search_of_a_whole_set)                          // src/glvq/glvq.rs:171-194, prototypes held fixed
{
    GLVQ model = GLVQ::from_fitted(lvq_prototypes());
    const Names labels = {"C0", "C0", "C1", "C1", "C1", "nope"};           // sample 2 is mislabelled, sample 5's label is unknown
    const auto e = model.evaluate(TOY, labels);
    CHECK((e.j_plus == std::vector<uint32_t>{0, 0, 1, 1, 1, 0}) && (e.j_minus == std::vector<uint32_t>{1, 1, 0, 0, 0, 1}));
    // sample 0 = (5,2): d+ = 1 to C0 (5,3), d- = sqrt(100 + 81) to C1 (-5,11)
    const double far0 = std::sqrt(181.0);
    CHECK(std::fabs(e.d_plus[0] - 1.0) < 1e-6 && std::fabs(e.d_minus[0] - far0) < 1e-5);
    CHECK(std::fabs(e.mu[0] - (1.0 - far0) / (1.0 + far0)) < 1e-6);
    CHECK(std::fabs(e.g_plus[0] - 4.0 * far0 / ((1.0 + far0) * (1.0 + far0))) < 1e-6);      // glvq.rs:193
    CHECK(std::fabs(e.g_minus[0] - 4.0 * 1.0 / ((1.0 + far0) * (1.0 + far0))) < 1e-6);      // glvq.rs:194
    CHECK(e.d_plus[1] == 0.0 && std::fabs(e.mu[1] + 1.0) < 1e-12);          // (5,3) sits on its prototype: mu = -1
    CHECK(e.mu[2] > 0.0 && e.misclassified == 1);                           // (5,4) labelled C1: nearer to the other class
    CHECK(std::isinf(e.d_plus[5]) && e.mu[5] == 0.0);                       // no prototype carries "nope": index 0, +INF, left out
    double want = 0.0;
    for (size_t i = 0; i < 5; ++i) want += e.mu[i];
    CHECK(std::fabs(e.cost - want) < 1e-12);
    should_panic("The data vector does not match the label vector in length.", [&] { model.evaluate(TOY, {"C0"}); });
    // with local Omegas the distances are the squared quadratic forms (src/lgmlvq/lgmlvq.rs:194-206)
    LGMLVQ local = LGMLVQ::from_fitted(lvq_prototypes(), {identity_over_sqrt_n(2), identity_over_sqrt_n(2)});
    const auto l = local.evaluate(TOY, {"C0", "C0", "C0", "C1", "C1", "C1"});
    CHECK(std::fabs(l.d_plus[0] - 0.5) < 1e-6 && std::fabs(l.d_minus[0] - 90.5) < 1e-4 && l.misclassified == 0);
}
TEST(glvq_batch_gradient_of_the_toy_set)                                   // glvq.rs:186-194 summed with the prototypes held fixed
{
    GLVQ model = GLVQ::from_fitted(lvq_prototypes());                       // C0 = (5,3), C1 = (-5,11)
    const Names labels = {"C0", "C0", "C1", "C1", "C1", "nope"};
    const auto e = model.evaluate(TOY, labels);
    const auto g = model.batch_gradient(TOY, labels);
    CHECK(g.used == 5 && std::fabs(g.cost - e.cost) < 1e-5);                // "nope" has no prototype: left out
    // the same sums in f64 from the evaluation's own factors
    const double W[2][2] = {{5.0, 3.0}, {-5.0, 11.0}};
    double want[2][2] = {{0, 0}, {0, 0}};
    for (size_t i = 0; i < 5; ++i)
        for (int d = 0; d < 2; ++d) {
            want[e.j_plus[i]][d] += -e.g_plus[i] * (TOY[i][d] - W[e.j_plus[i]][d]);        // glvq.rs:193
            want[e.j_minus[i]][d] += e.g_minus[i] * (TOY[i][d] - W[e.j_minus[i]][d]);      // glvq.rs:194
        }
    for (int k = 0; k < 2; ++k) CHECK(vec_all_close(g.dW[k], {want[k][0], want[k][1]}, 1e-4));
    should_panic("The data vector does not match the label vector in length.", [&] { model.batch_gradient(TOY, {"C0"}); });
}
TEST(ties_go_to_the_first_prototype)                                       // src/helpers.rs:70
{
    GLVQ model = GLVQ::from_fitted({{{1.0, 1.0}, "first"}, {{1.0, 1.0}, "second"}, {{9.0, 9.0}, "third"}});
    CHECK(model.predict(Data{{0.0, 0.0}, {1.0, 1.0}}) == (Names{"first", "first"}));
}
TEST(nan_sample_falls_back_to_index_zero)                                  // src/helpers.rs:58-59,70
{
    GLVQ model = GLVQ::from_fitted(lvq_prototypes());
    CHECK(model.predict(Data{{std::nan(""), 1.0}, {-5.0, 11.0}}) == (Names{"C0", "C1"}));
}
TEST(large_batch_through_the_packer)                                       // 70k separately allocated rows, 2-d
{
    GLVQ model = GLVQ::from_fitted(lvq_prototypes());
    Data data;
    for (int i = 0; i < 70000; ++i) data.push_back(TOY[i % 6]);
    const Names got = model.predict(data);
    bool ok = got.size() == data.size();
    for (size_t i = 0; ok && i < got.size(); ++i) ok = got[i] == (i % 6 < 3 ? "C0" : "C1");
    CHECK(ok);
}

int main(int argc, char **argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "--list")) {
        for (const Test &t : tests()) std::printf("%s\n", t.name);
        return 0;
    }
    int ran = 0, failed = 0;
    for (const Test &t : tests()) {
        if (argc > 1 && std::strcmp(argv[1], t.name)) continue;
        const int before = g_failures;
        std::printf("test %s ...\n", t.name);
        try {
            t.fn();
        } catch (const std::exception &e) {
            std::printf("    unexpected panic: %s\n", e.what());
            ++g_failures;
        }
        ++ran;
        if (g_failures != before) { ++failed; std::printf("test %s FAILED\n", t.name); }
    }
    std::printf("%d tests, %d failed\n", ran, failed);
    return failed ? 1 : 0;
}
