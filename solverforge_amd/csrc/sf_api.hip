// C ABI of the MI355X hot path (include/solverforge_amd.h): context, schema upload into
// SoA HBM tables, constraint/selector wiring, and kernel launches on the context's stream.
// No CPU fallback lives here: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define SF_TU_MAIN 1  // this unit compiles the plain kernels; the fused search kernels live in sf_tu_*.hip (sf_launch.h)
#include "../../include/solverforge_amd.h"
#include "sf_launch.h"
#include "sf_construct.hip"
#include "sf_clarke_wright.hip"
#include "sf_precedence.hip"

using namespace sf;

static std::string g_create_error;

struct Fact {
    int type = 0;  // 1 matrix i64, 2 col i32, 3 col u32, 4 csr u32
    void* d0 = nullptr;
    void* d1 = nullptr;
    int rows = 0, cols = 0;
    int64_t max_finite = 0;
    bool symmetric = false;  // matrices: data[i][j] == data[j][i] everywhere
    bool all_finite = false; // matrices: no negative / UNREACHABLE entry
    int64_t abs_sum = 0;     // i32 columns: sum of |value| (saturating at 2^62)
};
struct ConstraintSpec {
    int kind, desc, var, fact;
    int64_t param;
    int level;
    int64_t weight;
    std::vector<sf_pair_term> terms;  // SF_C_PAIR_JOIN_: the predicate program (sf_constraint_add_pair_join)
    std::vector<sf_uni_term> uterms;  // SF_C_UNI_PROGRAM_: filter program and weight expression (sf_constraint_add_uni_program)
    sf_uni_weight uweight{0, -1, -1, -1};
};
constexpr int SF_C_PAIR_JOIN_ = 100;  // internal kind of sf_constraint_add_pair_join
constexpr int SF_C_UNI_PROGRAM_ = 101;  // internal kind of sf_constraint_add_uni_program
// host copies of the components folded into ScalarModel::cost when a class carries uni programs (sf_evaluate_each rows): scaled cost and filter
struct UniComponent {
    size_t constraint_index;
    std::vector<int64_t> cost;
    std::vector<uint8_t> pass;
};
struct SelectorSpec {
    int kind, desc, var, max_nearby, fact;
    int min_size = 1, max_size = 3;  // sublist leaves; ruin leaf: min / max ruin count
    int moves_per_step = 10, max_source_len = 0, skip_empty = 0;  // ruin leaf (ListRuinMoveSelectorConfig)
    std::string variable_name;                                    // ruin leaf: scoped_seed hashes the variable name
};
struct NearbyScalarSource {  // sf_selector_add_nearby_scalar: rows ranked by (distance, source order, candidate)
    bool present = false;
    std::vector<uint32_t> off;
    std::vector<int32_t> val;
    uint32_t* d_off = nullptr;
    int32_t* d_val = nullptr;
};
struct PrecSpec {  // sf_constraint_add_list_precedence: the constraint's hooks as data
    bool on = false;
    int desc = 0, var = 0, hard_level = 0, mk_level = 1;
    std::vector<int32_t> dur, owner;
    std::vector<uint32_t> succ_off, succ;
    bool has_owner = false;
};
struct ClassSpec {
    int n_rows = 0;
    bool has_scalar = false;
    int var_index = 0, n_values = 0, allows_unassigned = 0;
    std::vector<int32_t> scalar_init;
    std::vector<uint32_t> value_off;  // ValueSource::EntitySlice: per-entity value lists (empty = the range 0..n_values)
    std::vector<int32_t> value_list;
    bool has_list = false;
    std::vector<uint32_t> list_off, list_vals;
    int element_capacity = 0, element_bound = 0;
};

struct sf_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int levels = 2, hard_levels = 1, R = 1;
    std::string err;
    std::map<int, ClassSpec> classes;
    std::map<int, Fact> facts;
    std::vector<ConstraintSpec> constraints;
    std::vector<SelectorSpec> selectors;
    bool initialized = false;
    // list model
    bool has_list_model = false;
    int list_desc = -1;
    ListModel lm{};
    NearbyScalarSource nearby_scalar[2];  // 0 = nearby value candidates (nearby change leaf), 1 = nearby entity candidates (nearby swap leaf)
    int nearby_scalar_dynamic = 0;
    PrecSpec prec;          // ListPrecedenceMakespanConstraint on the list class (sf_precedence.h)
    PrecModel pm{};
    bool prec_policy = false;  // sf_list_set_precedence_policy
    PlfModel plf{};  // critical-path precedence leaf: per-replica tables (allocated at the first launch that has the leaf)
    NbrIndex nbr{nullptr};  // presorted neighbour index (wave engine)
    ListModel lm_wave{};          // the list model as the COMPACT wave kernel sees it when the nodes are renumbered (ListModel::perm)
    NbrIndex nbr_wave{nullptr};   // ... and its neighbour index
    bool wave_renumbered = false;
    int last_wave_mode = -1;      // launch mode of the last wave-engine launch (sf_list_wave_layout)
    int xown_level = -1;             // SF_C_CROSS_OWNER_MATCH of a mixed model: level / weight / the [R][n_scalar] entity -> holding list map
    int64_t xown_weight = 0;
    std::vector<UniComponent> uni_components;  // non-empty: ScalarModel::cost is the fold of these (uni programs + at most one SF_C_VALUE_COST)
    uint16_t* d_xown_tab = nullptr;
    int64_t* d_xown_delta = nullptr;  // sf_apply / sf_apply_compound: the join's delta of the move being committed (xown_price)
    std::vector<std::pair<int, std::string>> providers;  // host-side providers declared through sf_provider_declare
    uint32_t* d_node_tab32 = nullptr;  // [R][dim] node -> slot tables of the generic engine's FAST + ruin kernel (GLeaves::node_tab)
    struct WaveFix {  // what differs in lm_wave from lm (applied at launch: the per-replica state pointers of lm may be set later)
        const uint16_t *perm, *inv, *mat16;
        const int32_t* demand;
        int32_t depot;
    } lm_wave_fix{};
    bool lm_small = false;  // every trial delta of the list model fits 32-bit arithmetic (wave engine MODE 2)
    int engine = SF_ENGINE_AUTO;
    // scalar model
    bool has_scalar_model = false;
    int scalar_desc = -1;
    ScalarModel sm{};
    // search
    sf_solver_config cfg{SF_ACCEPT_LATE_ACCEPTANCE, 400, SF_FORAGER_ACCEPTED_COUNT, 256, 1, SF_ORDER_RANDOM, 0};
    SearchParams sp{};
    bool search_alloc = false;
    // SimulatedAnnealingCalibration::default + DEFAULT_* (simulated_annealing.rs:11-38); seed_set = false -> cfg.random_seed
    sf_annealing_config anneal{SF_ANNEAL_CALIBRATED, 0, 128, 0, {0, 0, 0, 0}, 0.999985, 1.0e-9, 0.80, 1.0, 0};
    bool anneal_seed_set = false;
    double dla_tolerance = 0.01;  // DiversifiedLateAcceptanceAcceptor::default (diversified_late_acceptance.rs:106-110)
    uint64_t* d_explicit = nullptr;
    int64_t n_explicit = 0;
    // trace buffers
    int32_t* d_trace_moves = nullptr;
    int64_t* d_trace_scores = nullptr;
    int32_t* d_trace_flags = nullptr;
    int64_t* d_trace_count = nullptr;
    int32_t* d_trace_applied = nullptr;
    int64_t trace_cap = 0;
    // scratch
    int64_t* d_each = nullptr;           // [R][SF_EACH_WORDS] evaluate_each aggregates
    uint64_t* d_kopt_scratch = nullptr;  // [R][n_cap] distance keys of long routes (distance-pruned 3-opt leaf)
    uint64_t* d_ruin_rng = nullptr;      // [R][4] per-solve SmallRng state of the list ruin leaf
    uint32_t* d_mixed_ring = nullptr;    // [R][GL][GRC][2] candidate rings of the generic engine (+ [R][GL][GRC] side bytes)
    uint8_t* d_mixed_ringx = nullptr;
    int32_t* d_mixed_ringd = nullptr;    // [R][GL][GRC][2] trial deltas beside the rings (the FAST instantiations' scoring stage; allocated on first use)
    int union_order = -1;                // sf_union_configure: -1 = the default policy's root union
    std::vector<int32_t> union_weights;  // per leaf in union order; empty = equal
    int64_t* d_scores_out = nullptr;
    int32_t* d_ok = nullptr;
    // profiling
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // launches since the last fold / sf_profile_solve
    double prof_ms = 0;                                      // folded totals (events are bounded, see fold_events)
    int64_t prof_launches = 0;
    std::vector<void*> allocs;
    void* rccl = nullptr;  // portfolio state (sf_portfolio.cpp part below)
};

// dynamic LDS a workgroup may request: 160 KiB per CU minus the 1 KiB of static LDS the search kernels declare
// for the SimulatedAnnealing acceptor state (sf_anneal.h)
static constexpr size_t SF_LDS_BUDGET = 160 * 1024 - 1024;

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            return SF_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

// hipSetDevice is per host thread: every entry point that touches HIP binds the calling thread to the context's
// device for the duration of the call and restores the previous device afterwards, so contexts on different GPUs can
// be driven from one process / from worker threads other than the creator (a Rust rayon worker, bench.py's RCCL thread).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const sf_ctx* ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

static int fail(sf_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

template <class T>
static int dalloc(sf_ctx* ctx, T** out, size_t n) {
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, n * sizeof(T) + 16));
    ctx->allocs.push_back(p);
    HIPCHK(ctx, hipMemsetAsync(p, 0, n * sizeof(T) + 16, ctx->stream));
    *out = (T*)p;
    return SF_OK;
}
template <class T>
static int upload(sf_ctx* ctx, T** out, const T* src, size_t n) {
    int rc = dalloc(ctx, out, n ? n : 1);
    if (rc) return rc;
    if (n) HIPCHK(ctx, hipMemcpyAsync(*out, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" {

int32_t sf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t sf_ctx_create(int32_t device_id, int32_t score_levels, int32_t hard_levels, int32_t n_replicas,
                      sf_ctx** out) {
    if (!out) return SF_ERR_INVALID;
    *out = nullptr;
    if (score_levels < 1 || score_levels > SF_MAX_LEVELS || hard_levels < 0 || hard_levels > score_levels ||
        n_replicas < 1) {
        g_create_error = "invalid score levels / replica count";
        return SF_ERR_INVALID;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0 || device_id < 0 || device_id >= n) {
        g_create_error = "no HIP device available (the HIP path has no CPU fallback)";
        return SF_ERR_NO_DEVICE;
    }
    sf_ctx* ctx = new sf_ctx();
    ctx->device = device_id;
    ctx->levels = score_levels;
    ctx->hard_levels = hard_levels;
    ctx->R = n_replicas;
    {
        DeviceGuard _dev(ctx);
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device_id || hipStreamCreate(&ctx->stream) != hipSuccess) {
            g_create_error = "hipSetDevice/hipStreamCreate failed";
            delete ctx;
            return SF_ERR_HIP;
        }
    }
    *out = ctx;
    return SF_OK;
}

int32_t sf_portfolio_destroy(sf_ctx* ctx);

void sf_ctx_destroy(sf_ctx* ctx) {
    if (!ctx) return;
    {
        DeviceGuard _dev(ctx);
        (void)hipStreamSynchronize(ctx->stream);
        (void)sf_portfolio_destroy(ctx);  // a communicator the caller did not tear down
        for (void* p : ctx->allocs) (void)hipFree(p);
        for (auto& ev : ctx->events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

const char* sf_last_error(const sf_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t sf_sync(sf_ctx* ctx) {
    DeviceGuard _dev(ctx);
    if (!ctx) return SF_ERR_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

// ---- schema --------------------------------------------------------------------------
int32_t sf_schema_add_entity_class(sf_ctx* ctx, int32_t d, int32_t n_rows) {
    if (!ctx || d < 0 || n_rows < 0) return fail(ctx, SF_ERR_INVALID, "bad entity class");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "schema is frozen after sf_initialize");
    ctx->classes[d].n_rows = n_rows;
    return SF_OK;
}

int32_t sf_schema_add_scalar_variable(sf_ctx* ctx, int32_t d, int32_t var, int32_t n_values,
                                      int32_t allows_unassigned, const int32_t* initial) {
    if (!ctx || !ctx->classes.count(d) || !initial || n_values < 0)
        return fail(ctx, SF_ERR_INVALID, "bad scalar variable");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "schema is frozen after sf_initialize");
    ClassSpec& c = ctx->classes[d];
    if (c.has_scalar) return fail(ctx, SF_ERR_UNSUPPORTED, "one scalar planning variable per class");
    c.has_scalar = true;
    c.var_index = var;
    c.n_values = n_values;
    c.allows_unassigned = allows_unassigned;
    c.scalar_init.assign(initial, initial + c.n_rows);
    return SF_OK;
}

// ValueSource::EntitySlice: the value list of every entity of a scalar variable
int32_t sf_schema_set_value_lists(sf_ctx* ctx, int32_t d, int32_t var, const uint32_t* offsets, const int32_t* values) {
    if (!ctx || !ctx->classes.count(d) || !offsets) return fail(ctx, SF_ERR_INVALID, "bad value lists");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "schema is frozen after sf_initialize");
    ClassSpec& c = ctx->classes[d];
    if (!c.has_scalar || c.var_index != var) return fail(ctx, SF_ERR_INVALID, "value lists need the scalar variable declared first");
    if (offsets[0] != 0) return fail(ctx, SF_ERR_INVALID, "value list offsets must start at 0");
    for (int i = 0; i < c.n_rows; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SF_ERR_INVALID, "value list offsets must not decrease");
    const uint32_t total = offsets[c.n_rows];
    if (total > 0 && !values) return fail(ctx, SF_ERR_INVALID, "value lists: values is NULL");
    for (uint32_t k = 0; k < total; ++k)
        if (values[k] < 0 || values[k] >= c.n_values) return fail(ctx, SF_ERR_INVALID, "value list entry outside 0..n_values");
    c.value_off.assign(offsets, offsets + c.n_rows + 1);
    c.value_list.assign(values, values + total);
    if (c.value_list.empty()) c.value_list.push_back(0);  // keep the upload non-empty
    return SF_OK;
}

int32_t sf_schema_add_list_variable(sf_ctx* ctx, int32_t d, const uint32_t* offsets, const uint32_t* values,
                                    int32_t element_capacity, int32_t element_id_bound) {
    if (!ctx || !ctx->classes.count(d) || !offsets) return fail(ctx, SF_ERR_INVALID, "bad list variable");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "schema is frozen after sf_initialize");
    if (element_capacity < 0 || element_id_bound < 0) return fail(ctx, SF_ERR_INVALID, "negative element capacity / id bound");
    ClassSpec& c = ctx->classes[d];
    if (offsets[0] != 0) return fail(ctx, SF_ERR_INVALID, "list offsets must start at 0");
    for (int i = 0; i < c.n_rows; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SF_ERR_INVALID, "list offsets must be non-decreasing");
    const uint32_t total = offsets[c.n_rows];
    if (total > (uint32_t)element_capacity) return fail(ctx, SF_ERR_INVALID, "element_capacity too small");
    if (total > 0 && !values) return fail(ctx, SF_ERR_INVALID, "list values missing");
    for (uint32_t t = 0; t < total; ++t)
        if (values[t] >= (uint32_t)element_id_bound) return fail(ctx, SF_ERR_INVALID, "element id out of bound");
    c.has_list = true;
    c.list_off.assign(offsets, offsets + c.n_rows + 1);
    c.list_vals.assign(values, values + total);
    c.element_capacity = element_capacity;
    c.element_bound = element_id_bound;
    return SF_OK;
}

int32_t sf_fact_matrix_i64(sf_ctx* ctx, int32_t id, int32_t rows, int32_t cols, const int64_t* data) {
    DeviceGuard _dev(ctx);
    if (!ctx || !data || rows <= 0 || cols <= 0) return fail(ctx, SF_ERR_INVALID, "bad matrix");
    Fact f;
    f.type = 1;
    f.rows = rows;
    f.cols = cols;
    int64_t mx = 0;
    f.all_finite = true;
    for (size_t i = 0; i < (size_t)rows * cols; ++i) {
        if (data[i] >= 0 && data[i] != INT64_MAX) {
            if (data[i] > mx) mx = data[i];
        } else
            f.all_finite = false;
    }
    f.max_finite = mx;
    f.symmetric = rows == cols;
    for (int32_t i = 0; i < rows && f.symmetric; ++i)
        for (int32_t j = i + 1; j < cols; ++j)
            if (data[(size_t)i * cols + j] != data[(size_t)j * cols + i]) {
                f.symmetric = false;
                break;
            }
    int64_t* d = nullptr;
    int rc = upload(ctx, &d, data, (size_t)rows * cols);
    if (rc) return rc;
    f.d0 = d;
    ctx->facts[id] = f;
    return SF_OK;
}
int32_t sf_fact_column_i32(sf_ctx* ctx, int32_t id, int32_t n, const int32_t* data) {
    DeviceGuard _dev(ctx);
    if (!ctx || !data || n < 0) return fail(ctx, SF_ERR_INVALID, "bad column");
    Fact f;
    f.type = 2;
    f.rows = n;
    for (int32_t i = 0; i < n; ++i) {
        const int64_t a = data[i] < 0 ? -(int64_t)data[i] : (int64_t)data[i];
        if (f.abs_sum < ((int64_t)1 << 62)) f.abs_sum += a;
    }
    int32_t* d = nullptr;
    int rc = upload(ctx, &d, data, (size_t)n);
    if (rc) return rc;
    f.d0 = d;
    ctx->facts[id] = f;
    return SF_OK;
}
int32_t sf_fact_column_u32(sf_ctx* ctx, int32_t id, int32_t n, const uint32_t* data) {
    DeviceGuard _dev(ctx);
    if (!ctx || !data || n < 0) return fail(ctx, SF_ERR_INVALID, "bad column");
    Fact f;
    f.type = 3;
    f.rows = n;
    uint32_t* d = nullptr;
    int rc = upload(ctx, &d, data, (size_t)n);
    if (rc) return rc;
    f.d0 = d;
    ctx->facts[id] = f;
    return SF_OK;
}
int32_t sf_fact_csr_u32(sf_ctx* ctx, int32_t id, int32_t n_rows, const uint32_t* offsets, const uint32_t* values) {
    DeviceGuard _dev(ctx);
    if (!ctx || !offsets || n_rows < 0) return fail(ctx, SF_ERR_INVALID, "bad csr");
    if (offsets[0] != 0) return fail(ctx, SF_ERR_INVALID, "csr offsets must start at 0");
    for (int32_t i = 0; i < n_rows; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SF_ERR_INVALID, "csr offsets must be non-decreasing");
    if (offsets[n_rows] > 0 && !values) return fail(ctx, SF_ERR_INVALID, "csr values missing");
    if (offsets[n_rows] > 0x7FFFFFFFu) return fail(ctx, SF_ERR_UNSUPPORTED, "csr with more than 2^31 - 1 entries");
    Fact f;
    f.type = 4;
    f.rows = n_rows;
    f.cols = (int)offsets[n_rows];
    uint32_t *d0 = nullptr, *d1 = nullptr;
    int rc = upload(ctx, &d0, offsets, (size_t)n_rows + 1);
    if (rc) return rc;
    rc = upload(ctx, &d1, values, (size_t)offsets[n_rows]);
    if (rc) return rc;
    f.d0 = d0;
    f.d1 = d1;
    ctx->facts[id] = f;
    return SF_OK;
}

int32_t sf_constraint_add(sf_ctx* ctx, int32_t kind, int32_t d, int32_t var, int32_t fact_a, int64_t param,
                          int32_t level, int64_t weight) {
    if (!ctx || level < 0 || level >= ctx->levels) return fail(ctx, SF_ERR_INVALID, "bad constraint level");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "constraints are frozen after sf_initialize");
    if (kind < SF_C_UNI_UNASSIGNED || (kind > SF_C_BALANCE_VALUE && kind != SF_C_RUNS_VALUE && kind != SF_C_COMPLEMENTED_VALUE_SUM && kind != SF_C_PRESENCE_VALUE && kind != SF_C_CROSS_OWNER_MATCH)) return fail(ctx, SF_ERR_UNSUPPORTED, "constraint kind");
    ctx->constraints.push_back({kind, d, var, fact_a, param, level, weight, {}});
    return SF_OK;
}

int32_t sf_constraint_add_pair_join(sf_ctx* ctx, int32_t d, int32_t var, const sf_pair_term* terms, int32_t n_terms, int32_t level, int64_t weight) {
    if (!ctx || level < 0 || level >= ctx->levels) return fail(ctx, SF_ERR_INVALID, "bad constraint level");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "constraints are frozen after sf_initialize");
    if (!terms || n_terms < 1) return fail(ctx, SF_ERR_INVALID, "pair join: empty predicate program");
    if (n_terms > 8) return fail(ctx, SF_ERR_UNSUPPORTED, "pair join: at most 8 terms");
    int32_t prev = -1;
    for (int32_t t = 0; t < n_terms; ++t) {
        const sf_pair_term& pt = terms[t];
        if (pt.op < SF_PAIR_VALUE_EQ || pt.op > SF_PAIR_VALUE_ABSDIFF_LE) return fail(ctx, SF_ERR_INVALID, "pair join: unknown op");
        if (pt.clause < 0 || pt.clause > 127 || pt.clause < prev) return fail(ctx, SF_ERR_INVALID, "pair join: clause ids must ascend (0..127)");
        prev = pt.clause;
        const bool needs_fact = pt.op != SF_PAIR_VALUE_EQ && pt.op != SF_PAIR_VALUE_NE && pt.op != SF_PAIR_VALUE_ABSDIFF_LE;
        if (needs_fact && pt.fact < 0) return fail(ctx, SF_ERR_INVALID, "pair join: the op needs a fact");
        if (pt.op == SF_PAIR_TABLE_NONZERO && pt.fact_b < 0) return fail(ctx, SF_ERR_INVALID, "pair join: the table op needs a key column (fact_b)");
    }
    ConstraintSpec cs{SF_C_PAIR_JOIN_, d, var, -1, 0, level, weight, {}};
    cs.terms.assign(terms, terms + n_terms);
    ctx->constraints.push_back(cs);
    return SF_OK;
}

int32_t sf_constraint_add_uni_program(sf_ctx* ctx, int32_t d, int32_t var, const sf_uni_term* terms, int32_t n_terms, const sf_uni_weight* weight, int32_t level,
                                      int64_t scale) {
    if (!ctx || level < 0 || level >= ctx->levels) return fail(ctx, SF_ERR_INVALID, "bad constraint level");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "constraints are frozen after sf_initialize");
    if (n_terms < 0 || (n_terms > 0 && !terms) || !weight) return fail(ctx, SF_ERR_INVALID, "uni program: missing terms / weight");
    if (n_terms > 16) return fail(ctx, SF_ERR_UNSUPPORTED, "uni program: at most 16 terms");
    auto operands_ok = [&](int32_t lhs, int32_t f, int32_t fb, int32_t fc) {
        switch (lhs) {
            case SF_UNI_ROW_COL:
            case SF_UNI_VALUE_COL: return f >= 0;
            case SF_UNI_VALUE: return true;
            case SF_UNI_COL_DIFF:
            case SF_UNI_COL_ABSDIFF: return f >= 0 && fb >= 0;
            case SF_UNI_TABLE: return fc >= 0;
            default: return false;
        }
    };
    int32_t prev = -1;
    for (int32_t t = 0; t < n_terms; ++t) {
        const sf_uni_term& ut = terms[t];
        if (ut.cmp < SF_UNI_EQ || ut.cmp > SF_UNI_GE) return fail(ctx, SF_ERR_INVALID, "uni program: unknown comparison");
        if (!operands_ok(ut.lhs, ut.fact, ut.fact_b, ut.fact_c)) return fail(ctx, SF_ERR_INVALID, "uni program: unknown operand kind or missing fact");
        if (ut.clause < 0 || ut.clause > 127 || ut.clause < prev) return fail(ctx, SF_ERR_INVALID, "uni program: clause ids must ascend (0..127)");
        prev = ut.clause;
    }
    if (weight->lhs != 0 && !operands_ok(weight->lhs, weight->fact, weight->fact_b, weight->fact_c))
        return fail(ctx, SF_ERR_INVALID, "uni program: unknown weight operand or missing fact");
    ConstraintSpec cs{SF_C_UNI_PROGRAM_, d, var, -1, 0, level, scale, {}};
    if (n_terms > 0) cs.uterms.assign(terms, terms + n_terms);
    cs.uweight = *weight;
    ctx->constraints.push_back(cs);
    return SF_OK;
}

int32_t sf_constraint_add_list_precedence(sf_ctx* ctx, int32_t d, int32_t var, int32_t node_count, const int32_t* durations,
                                          const uint32_t* succ_offsets, const uint32_t* succ_values, const int32_t* expected_owner,
                                          int32_t hard_level, int32_t makespan_level) {
    if (!ctx || hard_level < 0 || hard_level >= ctx->levels || makespan_level < 0 || makespan_level >= ctx->levels || hard_level == makespan_level)
        return fail(ctx, SF_ERR_INVALID, "list precedence: bad score levels");
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "constraints are frozen after sf_initialize");
    if (ctx->prec.on) return fail(ctx, SF_ERR_UNSUPPORTED, "one list precedence constraint per context");
    if (node_count < 0 || (node_count > 0 && (!durations || !succ_offsets))) return fail(ctx, SF_ERR_INVALID, "list precedence: missing node data");
    if (node_count > 0 && succ_offsets[0] != 0) return fail(ctx, SF_ERR_INVALID, "list precedence: successor offsets must start at 0");
    int64_t dsum = 0;
    for (int32_t i = 0; i < node_count; ++i) {
        if (durations[i] < 0) return fail(ctx, SF_ERR_INVALID, "list precedence: negative duration");  // node_duration returns usize
        dsum += durations[i];
        if (succ_offsets[i + 1] < succ_offsets[i]) return fail(ctx, SF_ERR_INVALID, "list precedence: successor offsets must be non-decreasing");
    }
    if (dsum >= ((int64_t)1 << 31)) return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence: sum of durations >= 2^31 (32-bit earliest starts)");
    if (node_count > 0 && succ_offsets[node_count] > 0 && !succ_values) return fail(ctx, SF_ERR_INVALID, "list precedence: successor values missing");
    PrecSpec& ps = ctx->prec;
    ps.on = true;
    ps.desc = d;
    ps.var = var;
    ps.hard_level = hard_level;
    ps.mk_level = makespan_level;
    ps.dur.assign(durations, durations + node_count);
    ps.succ_off.assign(succ_offsets, succ_offsets + (node_count > 0 ? node_count + 1 : 0));
    if (node_count == 0) ps.succ_off.assign(1, 0u);
    ps.succ.assign(succ_values, succ_values + ps.succ_off.back());
    ps.has_owner = expected_owner != nullptr;
    if (expected_owner) ps.owner.assign(expected_owner, expected_owner + node_count);
    ctx->constraints.push_back({SF_C_LIST_PRECEDENCE_MAKESPAN, d, var, -1, 0, hard_level, (int64_t)makespan_level, {}});
    return SF_OK;
}

int32_t sf_selector_add(sf_ctx* ctx, int32_t kind, int32_t d, int32_t var, int32_t max_nearby, int32_t fact_meter) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    ctx->selectors.push_back({kind, d, var, max_nearby, fact_meter});
    return SF_OK;
}

// Nearby scalar leaves (NearbyChangeMoveSelector / NearbySwapMoveSelector of a scalar slot: scalar_neighborhood/cursor/change.rs:123-392,
// cursor/swap.rs:162-414).  The slot's nearby source and distance meter arrive as data; the rows are ranked here once
// (NearbyTopK's order: distance by f64::total_cmp, then source order, then candidate; non-finite distances dropped), so the device
// only applies the state-dependent filter and takes the first max_nearby survivors.
int32_t sf_selector_add_nearby_scalar(sf_ctx* ctx, int32_t kind, int32_t d, int32_t var, int32_t max_nearby, int64_t source_limit,
                                      const uint32_t* offsets, const int32_t* candidates, const double* distances, int32_t dynamic_slot) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    if (kind != SF_SEL_NEARBY_SCALAR_CHANGE && kind != SF_SEL_NEARBY_SCALAR_SWAP) return fail(ctx, SF_ERR_UNSUPPORTED, "nearby scalar selector kind");
    if (max_nearby < 1 || max_nearby > 63) return fail(ctx, SF_ERR_UNSUPPORTED, "nearby scalar leaves: max_nearby must be 1..63");
    if (!ctx->classes.count(d) || !ctx->classes[d].has_scalar) return fail(ctx, SF_ERR_INVALID, "nearby scalar leaf: the class has no scalar variable");
    if (!offsets || !candidates) return fail(ctx, SF_ERR_INVALID, "nearby scalar leaf: the nearby source rows are required (a slot without the hook passes its "
                                                                "ordinary candidate values / every entity)");
    const int which = kind == SF_SEL_NEARBY_SCALAR_CHANGE ? 0 : 1;
    NearbyScalarSource& src = ctx->nearby_scalar[which];
    if (src.present) return fail(ctx, SF_ERR_UNSUPPORTED, "one nearby scalar leaf of each kind per model");
    const ClassSpec& c = ctx->classes[d];
    const int n = c.n_rows;
    // built into locals and moved into the context on success only: a refused call leaves no partial rows behind for a retry
    std::vector<uint32_t> off_new(1, 0u);
    std::vector<int32_t> val_new;
    struct Ranked {
        double dist;
        uint32_t order;
        int32_t cand;
    };
    auto total_key = [](double v) {  // f64::total_cmp
        int64_t x;
        std::memcpy(&x, &v, 8);
        return x ^ (int64_t)((uint64_t)(x >> 63) >> 1);
    };
    std::vector<Ranked> row;
    for (int e = 0; e < n; ++e) {
        if (offsets[e + 1] < offsets[e]) return fail(ctx, SF_ERR_INVALID, "nearby scalar leaf: offsets must be non-decreasing");
        row.clear();
        // the source is visited with a limit: value_candidate_limit for values (change.rs:351-355), entity_count for entities (swap.rs:389)
        const uint64_t lim = which == 0 ? (source_limit > 0 ? (uint64_t)source_limit : ~0ull) : (uint64_t)n;
        for (uint32_t k = offsets[e]; k < offsets[e + 1] && (uint64_t)(k - offsets[e]) < lim; ++k) {
            const int32_t cand = candidates[k];
            // values must name a value of the variable; an entity candidate >= the entity count is legal input -- the reference's swap cursor
            // skips it when the row is filtered (cursor/swap.rs:346-398) and so does the device (`(uint32_t)c < ns`)
            if (cand < 0 || (which == 0 && cand >= c.n_values)) return fail(ctx, SF_ERR_INVALID, "nearby scalar leaf: candidate out of range");
            const uint32_t order = k - offsets[e];
            const double dist = distances ? distances[k] : (double)order;  // meter None: the source order (change.rs:332-334)
            if (!std::isfinite(dist)) continue;                            // NearbyTopK::push drops non-finite distances
            if (which == 1 && cand > 65535) return fail(ctx, SF_ERR_UNSUPPORTED, "nearby scalar swap: entity ids up to 65535");
            row.push_back({dist, order, cand});
        }
        std::sort(row.begin(), row.end(), [&](const Ranked& l, const Ranked& r) {
            const int64_t a = total_key(l.dist), b = total_key(r.dist);
            if (a != b) return a < b;
            if (l.order != r.order) return l.order < r.order;
            return l.cand < r.cand;
        });
        for (auto& x : row) val_new.push_back(x.cand);
        off_new.push_back((uint32_t)val_new.size());
    }
    src.off = std::move(off_new);
    src.val = std::move(val_new);
    src.present = true;
    ctx->nearby_scalar_dynamic = dynamic_slot ? 1 : 0;
    SelectorSpec s{kind, d, var, max_nearby, -1};
    ctx->selectors.push_back(s);
    return SF_OK;
}

int32_t sf_selector_add_kopt(sf_ctx* ctx, int32_t d, int32_t var, int32_t k, int32_t min_segment_len, int32_t max_nearby) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    if (k != 3) return fail(ctx, SF_ERR_UNSUPPORTED, "k-opt leaf: only k = 3 (the reference default) runs on the device");
    if (min_segment_len < 1 || min_segment_len > 4096) return fail(ctx, SF_ERR_INVALID, "k-opt min_segment_len must be >= 1");
    if (max_nearby < 0 || max_nearby > (int32_t)KOPT_MAX_NEARBY) return fail(ctx, SF_ERR_UNSUPPORTED, "k-opt max_nearby must be 0 (full enumeration) or 1..64");
    SelectorSpec s{SF_SEL_KOPT, d, var, max_nearby, -1};
    s.min_size = min_segment_len;
    s.max_size = min_segment_len;
    ctx->selectors.push_back(s);
    return SF_OK;
}

// ListPermuteMoveSelectorConfig (solverforge-config/src/move_selector.rs:391-416: windows of 2..=5 elements by default)
int32_t sf_selector_add_permute(sf_ctx* ctx, int32_t d, int32_t var, int32_t min_window_size, int32_t max_window_size) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    if (min_window_size < 2 || max_window_size < min_window_size || max_window_size > 8)  // list_leaf/spec.rs:260-265,325
        return fail(ctx, SF_ERR_INVALID, "list permute bounds require 2 <= min <= max <= 8");
    SelectorSpec s{SF_SEL_LIST_PERMUTE, d, var, 0, -1};
    s.min_size = min_window_size;
    s.max_size = max_window_size;
    ctx->selectors.push_back(s);
    return SF_OK;
}

// ListPrecedenceMoveSelector (heuristic/selector/list_precedence.rs:121-210; ListPrecedenceMoveConfig has no tunables): the critical-path leaf of
// a list class that carries the precedence constraint, whose fixed successors and durations are the leaf's hooks
int32_t sf_selector_add_precedence(sf_ctx* ctx, int32_t d, int32_t var) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    for (auto& s : ctx->selectors)
        if (s.kind == SF_SEL_LIST_PRECEDENCE) return fail(ctx, SF_ERR_UNSUPPORTED, "one list precedence leaf per union");
    ctx->selectors.push_back(SelectorSpec{SF_SEL_LIST_PRECEDENCE, d, var, 0, -1});
    return SF_OK;
}

// The compiled runtime slot's precedence policy (list_leaf/cursor/slot.rs:191-404, ruin_access.rs:195-217): the list slot declares its
// precedence successors to EVERY list leaf, not only to the critical-path one
int32_t sf_list_set_precedence_policy(sf_ctx* ctx, int32_t d, int32_t var, int32_t enabled) {
    if (!ctx) return SF_ERR_INVALID;
    (void)d, (void)var;
    ctx->prec_policy = enabled != 0;
    return SF_OK;
}

int32_t sf_selector_add_sublist(sf_ctx* ctx, int32_t kind, int32_t d, int32_t var, int32_t min_size, int32_t max_size) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    if (kind != SF_SEL_SUBLIST_CHANGE && kind != SF_SEL_SUBLIST_SWAP) return fail(ctx, SF_ERR_UNSUPPORTED, "sublist selector kind");
    if (min_size < 1 || max_size < min_size || max_size > 15) return fail(ctx, SF_ERR_INVALID, "sublist sizes must satisfy 1 <= min <= max <= 15");
    SelectorSpec s{kind, d, var, 0, -1};
    s.min_size = min_size;
    s.max_size = max_size;
    ctx->selectors.push_back(s);
    return SF_OK;
}

// UnionMoveSelectorConfig { selection_order, weighting } of the root union
int32_t sf_union_configure(sf_ctx* ctx, int32_t selection_order, const int64_t* weights, int32_t n_weights) {
    if (!ctx) return SF_ERR_INVALID;
    if (selection_order < -1 || selection_order > SF_UNION_STRATIFIED_RANDOM) return fail(ctx, SF_ERR_INVALID, "union selection order");
    if (n_weights < 0 || (n_weights > 0 && !weights)) return fail(ctx, SF_ERR_INVALID, "bad union weights");
    std::vector<int32_t> w;
    bool unit = true;
    for (int32_t i = 0; i < n_weights; ++i) {
        if (weights[i] < 0) return fail(ctx, SF_ERR_INVALID, "union weights are unsigned");
        if (weights[i] > 65535) return fail(ctx, SF_ERR_UNSUPPORTED, "union weights above 65535 are not supported on the device");
        unit = unit && weights[i] == 1;
        w.push_back((int32_t)weights[i]);
    }
    // UnionScheduler::new asserts this (vec_union.rs:215-222)
    if (!unit && selection_order != SF_UNION_RANDOM && selection_order != SF_UNION_STRATIFIED_RANDOM && selection_order != -1)
        return fail(ctx, SF_ERR_INVALID, "union weights require random or stratified_random selection order");
    ctx->union_order = selection_order;
    ctx->union_weights = unit ? std::vector<int32_t>() : w;
    return SF_OK;
}
static bool union_is_custom(const sf_ctx* ctx) {
    return (ctx->union_order >= 0 && ctx->union_order != SF_UNION_STRATIFIED_RANDOM) || !ctx->union_weights.empty();
}

// list ruin leaf (ListRuinMoveSelectorConfig, solverforge-config/src/move_selector.rs:552-587)
int32_t sf_selector_add_ruin(sf_ctx* ctx, int32_t d, int32_t var, int32_t min_ruin_count, int32_t max_ruin_count, int32_t moves_per_step,
                             int32_t max_source_list_len, int32_t skip_empty_destinations, const char* variable_name) {
    if (!ctx) return SF_ERR_INVALID;
    if (ctx->initialized) return fail(ctx, SF_ERR_INVALID, "selectors are frozen after sf_initialize");
    if (min_ruin_count < 1 || max_ruin_count < min_ruin_count) return fail(ctx, SF_ERR_INVALID, "ruin counts must satisfy 1 <= min <= max");
    if (max_ruin_count > (int32_t)RUIN_MAX_COUNT) return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin: at most 6 elements per move on the device");
    if (moves_per_step < 0 || max_source_list_len < 0) return fail(ctx, SF_ERR_INVALID, "list ruin: negative moves_per_step / max_source_list_len");
    if (moves_per_step > (int32_t)RUIN_MAX_MOVES) return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin: at most 16 moves per step on the device");
    SelectorSpec s{SF_SEL_LIST_RUIN, d, var, 0, -1};
    s.min_size = min_ruin_count;
    s.max_size = max_ruin_count;
    s.moves_per_step = moves_per_step;
    s.max_source_len = max_source_list_len;
    s.skip_empty = skip_empty_destinations != 0;
    s.variable_name = variable_name ? variable_name : "";
    ctx->selectors.push_back(s);
    return SF_OK;
}

}  // extern "C"

// std::hash::DefaultHasher = SipHash-1-3 with a zero key (published algorithm, Aumasson & Bernstein): `str::hash` feeds the bytes
// and a 0xFF terminator (heuristic/move/metadata.rs:125-129).  Host side only: it seeds the list ruin leaf's stream.
static uint64_t sip13_str(const std::string& str) {
    uint64_t v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL, v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    auto round = [&]() {
        v0 += v1, v1 = rotl(v1, 13), v1 ^= v0, v0 = rotl(v0, 32);
        v2 += v3, v3 = rotl(v3, 16), v3 ^= v2;
        v0 += v3, v3 = rotl(v3, 21), v3 ^= v0;
        v2 += v1, v1 = rotl(v1, 17), v1 ^= v2, v2 = rotl(v2, 32);
    };
    std::string data = str;
    data.push_back((char)0xFF);
    const size_t len = data.size();
    size_t i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t m = 0;
        for (int b = 0; b < 8; ++b) m |= (uint64_t)(uint8_t)data[i + b] << (8 * b);
        v3 ^= m;
        round();
        v0 ^= m;
    }
    uint64_t last = (uint64_t)(len & 0xFF) << 56;
    for (size_t b = 0; i + b < len; ++b) last |= (uint64_t)(uint8_t)data[i + b] << (8 * b);
    v3 ^= last;
    round();
    v0 ^= last;
    v2 ^= 0xFF;
    round(), round(), round();
    return v0 ^ v1 ^ v2 ^ v3;
}
// scoped_seed (heuristic/selector/seed.rs:3-17)
static uint64_t scoped_seed(uint64_t base_seed, uint64_t descriptor_index, const std::string& variable_name, const char* selector_kind) {
    auto rotl = [](uint64_t x, int b) { return (x << b) | (x >> (64 - b)); };
    return splitmix64(base_seed ^ (descriptor_index * 0x9E3779B97F4A7C15ULL) ^ rotl(sip13_str(variable_name), 17) ^ rotl(sip13_str(selector_kind), 41));
}
static const SelectorSpec* ruin_selector(const sf_ctx* ctx) {
    for (auto& s : ctx->selectors)
        if (s.kind == SF_SEL_LIST_RUIN && ctx->has_list_model && s.desc == ctx->list_desc) return &s;
    return nullptr;
}
// RuntimeListNeighborhoodStreamState::new (list_leaf/cursor.rs:117-145): replica r's stream = SmallRng::seed_from_u64(
// scoped_seed(random_seed + r, descriptor, variable, "list_ruin_move_selector"))
static int ruin_phase_start(sf_ctx* ctx) {
    const SelectorSpec* rs = ruin_selector(ctx);
    if (!rs) return SF_OK;
    if (!ctx->d_ruin_rng) {
        int rc = dalloc(ctx, &ctx->d_ruin_rng, (size_t)ctx->R * 4);
        if (rc) return rc;
    }
    std::vector<uint64_t> st((size_t)ctx->R * 4);
    for (int r = 0; r < ctx->R; ++r) {
        uint64_t state = scoped_seed(ctx->cfg.random_seed + (uint64_t)r, (uint64_t)rs->desc, rs->variable_name, "list_ruin_move_selector");
        for (int i = 0; i < 4; ++i) {  // xoshiro256++ seed_from_u64: splitmix64 expansion
            state += 0x9E3779B97F4A7C15ULL;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            st[(size_t)r * 4 + i] = z ^ (z >> 31);
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_ruin_rng, st.data(), st.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

// ---- model assembly --------------------------------------------------------------------
static int build_list_model(sf_ctx* ctx, int d) {
    ClassSpec& c = ctx->classes[d];
    ListModel& m = ctx->lm;
    m = ListModel{};
    m.V = c.n_rows;
    m.n_cap = c.element_capacity;
    m.dim = c.element_bound;
    m.levels = ctx->levels;
    m.cap_level = m.dist_level = m.ne_level = -1;
    if (m.V > 32767) return fail(ctx, SF_ERR_UNSUPPORTED, "list owners > 32767");
    for (auto& cs : ctx->constraints) {
        if (cs.kind == SF_C_ROUTE_CAPACITY && cs.desc == d) {
            if (!ctx->facts.count(cs.fact) || ctx->facts[cs.fact].type != 2) return fail(ctx, SF_ERR_INVALID, "capacity needs an i32 demand column");
            if (ctx->facts[cs.fact].rows < m.dim) return fail(ctx, SF_ERR_INVALID, "demand column shorter than element id bound");
            m.demand = (const int32_t*)ctx->facts[cs.fact].d0;
            m.capacity = cs.param;
            m.cap_level = cs.level;
            m.cap_weight = cs.weight;
        } else if (cs.kind == SF_C_ROUTE_DISTANCE && cs.desc == d) {
            if (!ctx->facts.count(cs.fact) || ctx->facts[cs.fact].type != 1) return fail(ctx, SF_ERR_INVALID, "distance needs an i64 matrix");
            Fact& f = ctx->facts[cs.fact];
            if (f.rows != f.cols || f.rows < m.dim) return fail(ctx, SF_ERR_INVALID, "matrix smaller than element id bound");
            m.mat = (const int64_t*)f.d0;
            m.dim = f.rows;
            if (cs.param < 0 || cs.param >= (int64_t)f.rows) return fail(ctx, SF_ERR_INVALID, "depot node outside the matrix");
            m.depot = (int32_t)cs.param;
            m.dist_level = cs.level;
            m.dist_weight = cs.weight;
        } else if (cs.kind == SF_C_NOT_EXISTS_FLATTENED && cs.desc == d) {
            if (!ctx->facts.count(cs.fact) || ctx->facts[cs.fact].type != 3) return fail(ctx, SF_ERR_INVALID, "not-exists needs a u32 key column");
            m.ne_keys = (const uint32_t*)ctx->facts[cs.fact].d0;
            m.ne_n = ctx->facts[cs.fact].rows;
            m.ne_level = cs.level;
            m.ne_weight = cs.weight;
        }
    }
    // nearby meter must be the same matrix (MatrixDistanceMeter)
    const auto demand_rows_ok = [&]() {  // the matrix (distance constraint or meter) may raise m.dim after the capacity constraint was visited
        for (auto& cs : ctx->constraints)
            if (cs.kind == SF_C_ROUTE_CAPACITY && cs.desc == d && ctx->facts[cs.fact].rows < m.dim) return false;
        return true;
    };
    for (auto& s : ctx->selectors)
        if ((s.kind == SF_SEL_NEARBY_LIST_CHANGE || s.kind == SF_SEL_NEARBY_LIST_SWAP) && s.desc == d) {
            if (!ctx->facts.count(s.fact) || ctx->facts[s.fact].type != 1) return fail(ctx, SF_ERR_INVALID, "nearby selector needs an i64 matrix meter");
            Fact& f = ctx->facts[s.fact];
            if (f.max_finite >= MAX_PACKED_DISTANCE) return fail(ctx, SF_ERR_UNSUPPORTED, "matrix values >= 2^40 (packed top-k keys)");
            if (m.mat && m.mat != (const int64_t*)f.d0) return fail(ctx, SF_ERR_UNSUPPORTED, "meter matrix must be the distance matrix");
            if (!m.mat) {
                m.mat = (const int64_t*)f.d0;
                m.dim = f.rows;
            }
            if (s.max_nearby < 1 || s.max_nearby > 64) return fail(ctx, SF_ERR_UNSUPPORTED, "max_nearby must be 1..64");
        }
    if (m.dim > 65535 * 16) return fail(ctx, SF_ERR_UNSUPPORTED, "node id bound too large");
    if (!demand_rows_ok()) return fail(ctx, SF_ERR_INVALID, "demand column shorter than the distance matrix (every matrix node needs a demand row)");
    const int R = ctx->R;
    int rc;
    if ((rc = dalloc(ctx, &m.visits, (size_t)R * m.n_cap))) return rc;
    if ((rc = dalloc(ctx, &m.off, (size_t)R * (m.V + 1)))) return rc;
    if ((rc = dalloc(ctx, &m.load, (size_t)R * m.V))) return rc;
    if ((rc = dalloc(ctx, &m.score, (size_t)R * 4))) return rc;
    if ((rc = dalloc(ctx, &m.best_visits, (size_t)R * m.n_cap))) return rc;
    if ((rc = dalloc(ctx, &m.best_off, (size_t)R * (m.V + 1)))) return rc;
    if ((rc = dalloc(ctx, &m.best_score, (size_t)R * 4))) return rc;
    // replica 0 from the host, then the replicas doubled on the device: 2 + 2 log2(R) copies instead of 2 R (a 98,304-replica context issued 196,608
    // four-kilobyte uploads, profiles/r06m_bench_kernel_stats.csv)
    HIPCHK(ctx, hipMemcpyAsync(m.visits, c.list_vals.data(), c.list_vals.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(m.off, c.list_off.data(), c.list_off.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    for (size_t k = 1; k < (size_t)R; k *= 2) {
        const size_t nrep = std::min(k, (size_t)R - k);
        HIPCHK(ctx, hipMemcpyAsync(m.visits + k * m.n_cap, m.visits, nrep * m.n_cap * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(m.off + k * (m.V + 1), m.off, nrep * (m.V + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->has_list_model = true;
    ctx->list_desc = d;
    ctx->pm = PrecModel{};
    if (ctx->prec.on) {  // ListPrecedenceMakespanConstraint: fixed graph + per-replica scratch (sf_precedence.h)
        PrecSpec& ps = ctx->prec;
        if (ps.desc != d) return fail(ctx, SF_ERR_INVALID, "list precedence: descriptor is not the list class");
        const int n = (int)ps.dur.size();
        if (c.element_bound > n) return fail(ctx, SF_ERR_INVALID, "list precedence: element ids must be < node_count");
        std::vector<char> seen((size_t)n, 0);
        for (uint32_t x : c.list_vals) {
            if (seen[x]) return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence: an element appears in more than one list position");
            seen[x] = 1;
        }
        for (auto& sel : ctx->selectors)  // its recreate is scored by the precedence constraint alone (plf_ruin)
            // (a flattened not-exists with its uniform weight costs every insertion of a round the same and nothing once all are back)
            if (sel.kind == SF_SEL_LIST_RUIN && (m.dist_level >= 0 || m.cap_level >= 0))
                return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin leaf on a precedence model with distance / capacity constraints");
        PrecModel& pm = ctx->pm;
        pm.on = 1;
        pm.hard_level = ps.hard_level;
        pm.mk_level = ps.mk_level;
        pm.n = n;
        std::vector<uint32_t> soff((size_t)n + 1, 0), sval, poff((size_t)n + 1, 0), pval;
        std::vector<int32_t> indeg((size_t)n, 0);
        int64_t invalid = 0;
        for (int i = 0; i < n; ++i) {
            for (uint32_t t = ps.succ_off[i]; t < ps.succ_off[i + 1]; ++t) {
                const uint32_t to = ps.succ[t];
                if (to < (uint32_t)n) {
                    sval.push_back(to);
                    indeg[to] += 1;
                } else
                    ++invalid;  // invalid_fixed_edges (list_precedence.rs:78-84)
            }
            soff[i + 1] = (uint32_t)sval.size();
        }
        pm.const_penalty = invalid;
        pm.n_edges = (int32_t)sval.size();
        {  // fixed predecessors (incremental trial refresh)
            for (int i = 0; i < n; ++i) poff[(size_t)i + 1] = poff[(size_t)i] + (uint32_t)indeg[(size_t)i];
            pval.resize(sval.size());
            std::vector<uint32_t> fill(poff.begin(), poff.end() - 1);
            for (int i = 0; i < n; ++i)
                for (uint32_t t = soff[(size_t)i]; t < soff[(size_t)i + 1]; ++t) pval[fill[sval[t]]++] = (uint32_t)i;
        }
        int32_t* d_dur = nullptr;
        uint32_t *d_soff = nullptr, *d_sval = nullptr;
        int32_t *d_indeg = nullptr, *d_owner = nullptr;
        if ((rc = upload(ctx, &d_dur, ps.dur.data(), (size_t)n))) return rc;
        if ((rc = upload(ctx, &d_soff, soff.data(), soff.size()))) return rc;
        if ((rc = upload(ctx, &d_sval, sval.data(), sval.size()))) return rc;
        if ((rc = upload(ctx, &d_indeg, indeg.data(), (size_t)n))) return rc;
        if (ps.has_owner && (rc = upload(ctx, &d_owner, ps.owner.data(), (size_t)n))) return rc;
        pm.dur = d_dur;
        pm.succ_off = d_soff;
        pm.succ = d_sval;
        pm.indeg0 = d_indeg;
        pm.owner = ps.has_owner ? d_owner : nullptr;
        {  // node records of the Kahn rounds (prec_eval / prec_eval_grouped): one 8-byte load instead of three dependent ones
            if (n >= 0xFFFFFF) return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence: node ids are 24 bits in the node records");
            std::vector<uint32_t> nd((size_t)n * 2);
            for (int i = 0; i < n; ++i) {
                const uint32_t dg = soff[(size_t)i + 1] - soff[(size_t)i];
                nd[(size_t)i * 2] = (uint32_t)ps.dur[(size_t)i];
                nd[(size_t)i * 2 + 1] = ((dg < 255u ? dg : 255u) << 24) | (dg ? sval[soff[(size_t)i]] : 0xFFFFFFu);
            }
            uint32_t* d_nd = nullptr;
            if ((rc = upload(ctx, &d_nd, nd.data(), nd.size()))) return rc;
            pm.nd = d_nd;
        }
        const size_t words = (size_t)R * (n ? n : 1);
        if ((rc = dalloc(ctx, &pm.earliest, words))) return rc;
        if ((rc = dalloc(ctx, &pm.indeg, words))) return rc;
        if ((rc = dalloc(ctx, &pm.queue, words))) return rc;
        if ((rc = dalloc(ctx, &pm.lsucc, words))) return rc;
        if ((rc = dalloc(ctx, &pm.state, (size_t)R * 2))) return rc;
        uint32_t *d_poff = nullptr, *d_pval = nullptr;
        if ((rc = upload(ctx, &d_poff, poff.data(), poff.size()))) return rc;
        if ((rc = upload(ctx, &d_pval, pval.data(), pval.size()))) return rc;
        pm.pred_off = d_poff;
        pm.pred = d_pval;
        if ((rc = dalloc(ctx, &pm.lpred, words))) return rc;
        if ((rc = dalloc(ctx, &pm.stamp_e, words))) return rc;
        if ((rc = dalloc(ctx, &pm.stamp_q, words))) return rc;
        if ((rc = dalloc(ctx, &pm.changed, words))) return rc;
        if ((rc = dalloc(ctx, &pm.queue2, words))) return rc;
        if ((rc = dalloc(ctx, &pm.pos, words))) return rc;
        if ((rc = dalloc(ctx, &pm.rnd, words))) return rc;
        if ((rc = dalloc(ctx, &pm.rec, words * 16))) return rc;
        if ((rc = dalloc(ctx, &pm.roff, words + (size_t)R))) return rc;
        if ((rc = dalloc(ctx, &pm.pmax, words + (size_t)R))) return rc;
        pm.has_zero_duration = 0;
        {
            int64_t dsum = 0;
            for (int32_t dv : ps.dur) {
                if (dv <= 0) pm.has_zero_duration = 1;
                dsum += dv > 0 ? dv : 0;
            }
            if (dsum >= ((int64_t)1 << 27)) pm.has_zero_duration = 1;  // keep transient sweep values far from the i32 range
        }
    }
    // compact u32 matrix copy (4-byte gathers in the trial-score path) when every finite leg fits
    for (auto& kv : ctx->facts)
        if (kv.second.type == 1 && kv.second.d0 == (void*)m.mat && kv.second.max_finite < 0xFFFFFFFFLL) {
            uint32_t* m32 = nullptr;
            const size_t n = (size_t)m.dim * m.dim;
            if ((rc = dalloc(ctx, &m32, n))) return rc;
            hipLaunchKernelGGL(k_mat_compress, dim3(1024), dim3(256), 0, ctx->stream, m.mat, n, m32);
            HIPCHK(ctx, hipGetLastError());
            m.mat32 = m32;
            static const bool no16 = std::getenv("SF_AMD_NO_MAT16") != nullptr;  // diagnostics: A/B
            if (kv.second.max_finite < 0xFFFF && !no16) {  // half-size copy for the trial gathers (ListModel::mat16)
                uint16_t* m16 = nullptr;
                if ((rc = dalloc(ctx, &m16, n))) return rc;
                hipLaunchKernelGGL(k_mat_compress16, dim3(1024), dim3(256), 0, ctx->stream, m.mat, n, m16);
                HIPCHK(ctx, hipGetLastError());
                m.mat16 = m16;
            }
        }
    for (auto& kv : ctx->facts)
        if (kv.second.type == 1 && kv.second.d0 == (void*)m.mat) {
            m.mat_symmetric = kv.second.symmetric ? 1 : 0;
            m.leg16 = (kv.second.symmetric && m.mat32 && m.dist_level >= 0 && kv.second.max_finite < 0xFFFF && m.dim <= 65535) ? 1 : 0;
            const bool no16 = std::getenv("SF_AMD_NO_LEG16") != nullptr;  // diagnostics / parity tests: force the general ruin path
            if (no16) m.leg16 = 0;
        }
    {  // 32-bit trial arithmetic (k_list_search_wave MODE 2): every leg finite and < 2^26, |level delta| < 2^30
        const int64_t lim = (int64_t)1 << 28;
        auto mag = [](int64_t v) { return v < 0 ? (v == INT64_MIN ? INT64_MAX : -v) : v; };
        bool ok = m.mat32 != nullptr && m.dist_level >= 0;
        int64_t max_leg = 0, dem = 0;
        for (auto& kv : ctx->facts) {
            if (kv.second.type == 1 && kv.second.d0 == (void*)m.mat) {
                ok = ok && kv.second.all_finite && kv.second.max_finite < ((int64_t)1 << 26);
                max_leg = kv.second.max_finite;
            }
            if (kv.second.type == 2 && kv.second.d0 == (void*)m.demand) dem = kv.second.abs_sum;
        }
        ok = ok && mag(m.dist_weight) < lim && mag(m.dist_weight) * 8 * (max_leg + 1) < ((int64_t)1 << 29);
        if (m.cap_level >= 0)
            ok = ok && dem < lim && mag(m.capacity) < lim && mag(m.cap_weight) < lim &&
                 mag(m.cap_weight) * 2 * (dem + mag(m.capacity) + 1) < ((int64_t)1 << 29);
        {
            static const bool off = std::getenv("SF_AMD_NO_SMALL") != nullptr;  // diagnostics: A/B against MODE 1
            if (off) ok = false;
        }
        ctx->lm_small = ok;
        m.small32 = ok ? 1 : 0;
    }
    // presorted neighbour index for the wave engine: every matrix row sorted by (distance, node)
    bool nearby = false;
    for (auto& s : ctx->selectors)
        if ((s.kind == SF_SEL_NEARBY_LIST_CHANGE || s.kind == SF_SEL_NEARBY_LIST_SWAP) && s.desc == d) nearby = true;
    int P = 2;
    while (P < m.dim) P <<= 1;
    if (nearby && m.mat && (size_t)P * 8 <= 128 * 1024) {
        uint16_t* keys = nullptr;
        if ((rc = dalloc(ctx, &keys, (size_t)m.dim * m.dim))) return rc;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)k_nbr_presort, hipFuncAttributeMaxDynamicSharedMemorySize, P * 8));
        hipLaunchKernelGGL(k_nbr_presort, dim3(m.dim), dim3(256), (size_t)P * 8, ctx->stream, m.mat, m.dim, P, keys);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->nbr = NbrIndex{keys};
    }
    // Internal node numbering for the COMPACT wave kernel (ListModel::perm): once the u16 matrix and the neighbour index outgrow an
    // XCD's 4 MiB L2, every 2-byte leg gather pulls its own line from the Infinity Cache / HBM (CVRP-5000: 352 B of memory-side traffic
    // per candidate for <= 16 B of legs).  Nodes that are near each other get neighbouring ids -- a nearest-neighbour chain from the
    // depot, read off the presorted index -- so the legs of a trial (route neighbours, nearby destinations) sit a few entries off the
    // diagonal of their rows and the hot part of the matrix is a band that stays L2-resident.
    ctx->wave_renumbered = false;
    {
        const char* env = std::getenv("SF_AMD_RENUMBER");  // diagnostics / parity tests: 1 = always, 0 = never (read per model: a test can toggle it)
        const bool want = env ? std::atoi(env) != 0 : (size_t)m.dim * m.dim * 2 > (size_t)1024 * 1024;  // (CVRP-1000: 2 MB matrix + 2 MB index share a 4 MiB L2 with the replicas' state: +2 %; CVRP-5000: 352 -> 166 B of memory-side traffic per candidate)
        if (want && ctx->nbr.keys && m.mat16 && ctx->lm_small && m.dim <= 0x7FFF && m.depot >= 0 && m.depot < m.dim) {
            const int dim = m.dim;
            std::vector<uint16_t> hk((size_t)dim * dim);
            HIPCHK(ctx, hipMemcpyAsync(hk.data(), ctx->nbr.keys, hk.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            std::vector<uint16_t> perm((size_t)dim), inv((size_t)dim);
            std::vector<uint8_t> seen((size_t)dim, 0);
            std::vector<int> ptr((size_t)dim, 0);  // entries of a row before ptr are all visited: the scans are amortised over the chain
            int cur = m.depot, low = 0;
            for (int n = 0; n < dim; ++n) {
                perm[(size_t)cur] = (uint16_t)n;
                inv[(size_t)n] = (uint16_t)cur;
                seen[(size_t)cur] = 1;
                if (n + 1 == dim) break;
                int nx = -1;
                const uint16_t* row = hk.data() + (size_t)cur * dim;
                for (int& t = ptr[(size_t)cur]; t < dim; ++t) {
                    const uint32_t e = row[t];
                    if (e == 0xFFFFu) break;  // past the finite entries
                    if (!seen[e & 0x7FFFu]) {
                        nx = (int)(e & 0x7FFFu);
                        break;
                    }
                }
                if (nx < 0) {  // nothing reachable left from here: the lowest unvisited id
                    while (seen[(size_t)low]) ++low;
                    nx = low;
                }
                cur = nx;
            }
            uint16_t *d_perm = nullptr, *d_inv = nullptr, *d_m16 = nullptr, *d_keys = nullptr;
            int32_t* d_dem = nullptr;
            if ((rc = upload(ctx, &d_perm, perm.data(), perm.size()))) return rc;
            if ((rc = upload(ctx, &d_inv, inv.data(), inv.size()))) return rc;
            if ((rc = dalloc(ctx, &d_m16, (size_t)dim * dim))) return rc;
            if ((rc = dalloc(ctx, &d_keys, (size_t)dim * dim))) return rc;
            hipLaunchKernelGGL(k_mat16_renumber, dim3(dim), dim3(256), 0, ctx->stream, m.mat16, d_inv, dim, d_m16);
            hipLaunchKernelGGL(k_nbr_renumber, dim3(dim), dim3(256), 0, ctx->stream, ctx->nbr.keys, d_inv, d_perm, dim, d_keys);
            if (m.demand) {
                if ((rc = dalloc(ctx, &d_dem, (size_t)dim))) return rc;
                hipLaunchKernelGGL(k_i32_renumber, dim3((dim + 255) / 256), dim3(256), 0, ctx->stream, m.demand, d_inv, dim, d_dem);
            }
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            ctx->lm_wave_fix = {d_perm, d_inv, d_m16, d_dem, (int32_t)perm[(size_t)m.depot]};
            ctx->nbr_wave = NbrIndex{d_keys};
            ctx->wave_renumbered = true;
        }
    }
    return SF_OK;
}

// engine resolution: WAVE needs the neighbour index, u16-packable coordinates and an LDS slice per
// replica small enough for several replicas per CU.
static int list_max_nearby(sf_ctx* ctx) {
    int mk = 1;
    for (auto& s : ctx->selectors)
        if (s.desc == ctx->list_desc && (s.kind == SF_SEL_NEARBY_LIST_CHANGE || s.kind == SF_SEL_NEARBY_LIST_SWAP) &&
            s.max_nearby > mk)
            mk = s.max_nearby;
    return mk;
}
static bool wave_engine_possible(sf_ctx* ctx) {
    const ListModel& m = ctx->lm;
    // u16 element ids / ordinals in LDS; one wave's LDS slice must fit a CU
    if (!ctx->nbr.keys || m.dim > 16384 || m.n_cap + m.V > 65535 || list_max_nearby(ctx) > 64) return false;
    return WCarve(m.V, m.n_cap, m.dim, list_max_nearby(ctx)).total <= SF_LDS_BUDGET;
}
static bool use_wave_engine(sf_ctx* ctx) {
    if (ctx->engine == SF_ENGINE_BLOCK) return false;
    if (ctx->engine == SF_ENGINE_WAVE) return true;  // validated in sf_solver_set_engine / launch
    return wave_engine_possible(ctx) && WCarve(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim, list_max_nearby(ctx)).total <= SF_LDS_BUDGET / 2;
}

static int build_scalar_model(sf_ctx* ctx, int d);  // sf_api_scalar.inc
static size_t scalar_table_bytes(sf_ctx* ctx);
static int launch_scalar_search(sf_ctx* ctx, SearchParams& p, int grid, bool trace);

static int alloc_search(sf_ctx* ctx) {
    if (ctx->search_alloc) return SF_OK;
    SearchParams& p = ctx->sp;
    const int R = ctx->R;
    int rc;
    int la = ctx->cfg.late_acceptance_size > 0 ? ctx->cfg.late_acceptance_size : 1;
    if ((rc = dalloc(ctx, &p.last_step_score, (size_t)R * 4))) return rc;
    if ((rc = dalloc(ctx, &p.la_hist, (size_t)R * la * 4))) return rc;
    if ((rc = dalloc(ctx, &p.la_idx, (size_t)R))) return rc;
    if ((rc = dalloc(ctx, &p.step_index, (size_t)R))) return rc;
    if ((rc = dalloc(ctx, &p.seed_draws, (size_t)R))) return rc;
    if ((rc = dalloc(ctx, &p.stats, (size_t)R * SF_STATS_WORDS))) return rc;
    if ((rc = dalloc(ctx, &p.has_best, (size_t)R))) return rc;
    if ((rc = dalloc(ctx, &ctx->d_trace_count, 1))) return rc;
    if ((rc = dalloc(ctx, &ctx->d_trace_applied, 8))) return rc;
    if ((rc = dalloc(ctx, &ctx->d_ok, 4))) return rc;
    if ((rc = dalloc(ctx, &p.sa.state, (size_t)R * SA_WORDS))) return rc;
    if ((rc = dalloc(ctx, &p.dla_best, (size_t)R * 4))) return rc;
    p.la_size = la;
    ctx->search_alloc = true;
    return SF_OK;
}

static int ensure_trace(sf_ctx* ctx, int64_t cap) {
    if (cap <= ctx->trace_cap) return SF_OK;
    int rc;
    if ((rc = dalloc(ctx, &ctx->d_trace_moves, (size_t)cap * 6))) return rc;
    if ((rc = dalloc(ctx, &ctx->d_trace_scores, (size_t)cap * 4))) return rc;
    if ((rc = dalloc(ctx, &ctx->d_trace_flags, (size_t)cap))) return rc;
    ctx->trace_cap = cap;
    return SF_OK;
}

static void fill_search_params(sf_ctx* ctx, SearchParams& p) {
    p.acceptor = ctx->cfg.acceptor;
    p.dla_tolerance = ctx->dla_tolerance;
    p.forager = ctx->cfg.forager;
    p.limit = ctx->cfg.accepted_count_limit > 0 ? ctx->cfg.accepted_count_limit : 1;
    // FirstLastStepScoreImprovingForager: accepted_count_limit is an Option (improving.rs:128-137); <= 0 = None
    if (ctx->cfg.forager == SF_FORAGER_FIRST_LAST_STEP_SCORE_IMPROVING && ctx->cfg.accepted_count_limit <= 0) p.limit = 0;
    p.random_ties = ctx->cfg.random_ties;
    p.order = ctx->cfg.selection_order;
    p.random_seed = ctx->cfg.random_seed;
    p.dry_run = 0;
    {  // diagnostics only (A/B measurements of the generic engine's trial evaluators)
        static const bool legacy = std::getenv("SF_AMD_LEGACY_EVAL") != nullptr;
        p.legacy_eval = legacy ? 1 : 0;
    }
    p.replica_base = 0;
    p.explicit_seeds = ctx->d_explicit;
    p.n_explicit = ctx->n_explicit;
    p.trace_replica = -1;
    p.trace_moves = ctx->d_trace_moves;
    p.trace_scores = ctx->d_trace_scores;
    p.trace_flags = ctx->d_trace_flags;
    p.trace_cap = ctx->trace_cap;
    p.trace_count = ctx->d_trace_count;
    p.trace_applied = ctx->d_trace_applied;
}

static int fill_list_leaves(sf_ctx* ctx, SearchParams& p) {
    p.n_leaves = 0;
    // default-policy declaration order: nearby change, then nearby swap (policy/list.rs:24-33)
    for (int kind : {SF_SEL_NEARBY_LIST_CHANGE, SF_SEL_NEARBY_LIST_SWAP})
        for (auto& s : ctx->selectors)
            if (s.kind == kind && s.desc == ctx->list_desc) {
                if (p.n_leaves >= MAX_LEAVES) return fail(ctx, SF_ERR_UNSUPPORTED, "more than two list leaves");
                p.leaf[p.n_leaves++] = LeafSpec{s.kind, s.max_nearby, s.desc};
            }
    // (unions with any other leaf kind are routed to the generic N-leaf engine by launch_search)
    if (p.n_leaves == 0) return fail(ctx, SF_ERR_INVALID, "no list selector configured");
    return SF_OK;
}

static SearchLaunch make_launch(sf_ctx* ctx, const SearchParams* p, int grid, int block, size_t lds, const GLeaves* gl = nullptr) {
    return SearchLaunch{grid, block, lds, ctx->stream, &ctx->lm, &ctx->sm, gl, p, ctx->has_list_model ? 1 : 0, ctx->has_scalar_model ? 1 : 0, ctx->nbr};
}
template <int L>
static int launch_list_search_t(sf_ctx* ctx, const SearchParams& p, int grid, bool trace) {
    Carve<L> cv(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim);
    size_t lds = cv.total;
    if (lds > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "problem does not fit the 160 KiB LDS of one CU");
    HIPCHK(ctx, launch_tu_list_block<L>(trace, make_launch(ctx, &p, grid, 1024, lds)));
    return SF_OK;
}
template <int L>
static int launch_list_wave_t(sf_ctx* ctx, const SearchParams& p, int n_replicas, bool trace) {
    const bool fast = !trace && ctx->lm.mat32 && ctx->lm.dist_level >= 0 && p.acceptor == 1 && p.forager == 0 && !p.dry_run && p.n_leaves == 2 &&
                      p.leaf[0].kind == SF_SEL_NEARBY_LIST_CHANGE && p.leaf[1].kind == SF_SEL_NEARBY_LIST_SWAP && p.order == SF_ORDER_RANDOM;
    // replicas (waves) per workgroup: as many as the LDS holds, <= WPB; resident replicas per CU = whole workgroups in 160 KiB
    auto plan = [&](bool compact, size_t wave_cap, int& wpb_out, size_t& lds_out, bool node_global = false) {
        WCarve cvx(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim, list_max_nearby(ctx), compact, node_global);
        size_t best = 0;
        wpb_out = 1;
        const char* wenv = std::getenv("SF_AMD_WAVE_WPB");  // diagnostics: cap the replicas per workgroup (A/B of the workgroup shape)
        const int wmax = wenv && std::atoi(wenv) >= 1 && std::atoi(wenv) < WPB ? std::atoi(wenv) : WPB;
        for (int w = 1; w <= wmax; ++w) {  // the workgroup size that keeps the most replicas resident (a workgroup's LDS is allocated whole)
            const size_t per_wg = cvx.total * (size_t)w + (fast ? 0 : 1024);  // + the static annealing state (the FAST instantiations have none)
            if (cvx.total * (size_t)w > SF_LDS_BUDGET) break;
            size_t groups = (160 * 1024) / per_wg;
            if (groups * (size_t)w > wave_cap) groups = wave_cap / (size_t)w;  // waves per CU by register budget
            if (groups * (size_t)w >= best) {
                best = groups * (size_t)w;
                wpb_out = w;
            }
        }
        lds_out = cvx.total * (size_t)wpb_out;
        return best;
    };
    int wpb = 1;
    size_t lds = 0;
    const size_t resident_wide = plan(false, 4 * SF_WAVES_PER_EU, wpb, lds);
    int mode = fast ? (ctx->lm_small ? 2 : 1) : 0;
    if (mode == 2 && node_slot_compact_ok(ctx->lm.V) && ctx->lm.mat16) {  // (the COMPACT kernels gather from the u16 matrix)
        // the COMPACT slice when it puts more replicas on a CU: CVRP-5000 5 instead of 3 (LDS-bound, compiled for 4 waves per SIMD);
        // CVRP-1000 20 instead of 16 with the instantiation compiled for 5 waves per SIMD
        static const bool no_compact = std::getenv("SF_AMD_NO_COMPACT") != nullptr;  // diagnostics: A/B
        int wpb_c = 1;
        size_t lds_c = 0;
        static const int max_wpe = std::getenv("SF_AMD_WAVE_WPE") ? std::atoi(std::getenv("SF_AMD_WAVE_WPE")) : 6;  // diagnostics: cap the waves per SIMD (4 / 5 / 6)
        const size_t r6 = (no_compact || max_wpe < 6) ? 0 : plan(true, 24, wpb_c, lds_c);
        const size_t r5 = (no_compact || max_wpe < 5 || r6 > 20) ? 0 : plan(true, 20, wpb_c, lds_c);
        if (r6 > 20 && r6 > resident_wide) {  // 24 replicas per CU: the instantiation compiled for 6 waves per SIMD (80 VGPRs)
            mode = 5;
            wpb = wpb_c;
            lds = lds_c;
        } else if (r5 > 16 && r5 > resident_wide) {
            mode = 4;
            wpb = wpb_c;
            lds = lds_c;
        } else if (!no_compact && plan(true, 4 * SF_WAVES_PER_EU, wpb_c, lds_c) > resident_wide) {
            mode = 3;
            wpb = wpb_c;
            lds = lds_c;
        }
        // large models: the node -> slot table in HBM when that puts more replicas on a CU (a wave runs as fast at CVRP-5000 as at
        // CVRP-1000; the slice decides how many are resident: 29 KB = 5 per CU, 19 KB = 8)
        if (mode >= 3) {
            const char* ng = std::getenv("SF_AMD_NODE_GLOBAL");  // diagnostics / parity tests: 0 = never, 1 = whenever the COMPACT slice is taken
            const int ngv = ng ? std::atoi(ng) : -1;
            int wpb_g = 1, wpb_3 = 1;
            size_t lds_g = 0, lds_3 = 0;
            const size_t r3 = plan(true, 4 * SF_WAVES_PER_EU, wpb_3, lds_3);
            const size_t rg = plan(true, 4 * SF_WAVES_PER_EU, wpb_g, lds_g, true);
            // (round 5 also carried a 64-register / 32-replicas-per-CU instantiation, launch mode 7: parity-green and 1.5 % SLOWER at CVRP-1000 --
            // the scalar pipe is the bound, more waves do not help -- removed in round 6 with its untested code path)
            if (ngv != 0 && (ngv == 1 || (mode == 3 && rg > r3))) {
                if (!ctx->lm.node_tab) {
                    uint16_t* nt = nullptr;
                    int rc = dalloc(ctx, &nt, (size_t)ctx->R * ctx->lm.dim);
                    if (rc) return rc;
                    ctx->lm.node_tab = nt;
                }
                mode = 6;
                wpb = wpb_g;
                lds = lds_g;
            }
        }
    }
    SearchParams q = p;
    q.n_launch = n_replicas;
    SearchLaunch la = make_launch(ctx, &q, (n_replicas + wpb - 1) / wpb, 64 * wpb, lds);
    ctx->last_wave_mode = mode;
    if (mode >= 3 && ctx->wave_renumbered) {  // the COMPACT kernels on the internal node numbering (ListModel::perm)
        ctx->lm_wave = ctx->lm;
        ctx->lm_wave.perm = ctx->lm_wave_fix.perm, ctx->lm_wave.inv = ctx->lm_wave_fix.inv, ctx->lm_wave.mat16 = ctx->lm_wave_fix.mat16;
        if (ctx->lm.demand) ctx->lm_wave.demand = ctx->lm_wave_fix.demand;
        ctx->lm_wave.depot = ctx->lm_wave_fix.depot;
        la.lm = &ctx->lm_wave;
        la.nb = ctx->nbr_wave;
    }
    HIPCHK(ctx, launch_tu_list_wave<L>(trace, mode, la));
    return SF_OK;
}
static int launch_list_wave(sf_ctx* ctx, const SearchParams& p, int grid, bool trace) {
    if (!wave_engine_possible(ctx)) return fail(ctx, SF_ERR_UNSUPPORTED, "wave engine cannot run this model (needs a nearby matrix meter, <= 65535 elements, LDS slice <= 160 KiB)");
    // kernels are instantiated for 2 and 4 score levels; 1- and 3-level models run with one padded
    // (always zero) least-significant level, which never changes a lexicographic comparison
    if (ctx->levels <= 2) return launch_list_wave_t<2>(ctx, p, grid, trace);
    return launch_list_wave_t<4>(ctx, p, grid, trace);
}
static int launch_list_search(sf_ctx* ctx, const SearchParams& p, int grid, bool trace) {
    if (use_wave_engine(ctx)) return launch_list_wave(ctx, p, grid, trace);
    if (p.acceptor == SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE) return fail(ctx, SF_ERR_UNSUPPORTED, "DiversifiedLateAcceptance: wave, scalar and generic engines only");
    if (ctx->levels <= 2) return launch_list_search_t<2>(ctx, p, grid, trace);
    return launch_list_search_t<4>(ctx, p, grid, trace);
}

static int download_scores(sf_ctx* ctx, const int64_t* d_src4, int64_t* out) {
    std::vector<int64_t> tmp((size_t)ctx->R * 4);
    HIPCHK(ctx, hipMemcpyAsync(tmp.data(), d_src4, tmp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < ctx->R; ++r)
        for (int k = 0; k < ctx->levels; ++k) out[(size_t)r * ctx->levels + k] = tmp[(size_t)r * 4 + k];
    return SF_OK;
}

static int run_evaluate_all(sf_ctx* ctx, int64_t* out, int commit, int64_t* d_parts = nullptr) {
    int rc;
    if (!ctx->d_scores_out && (rc = dalloc(ctx, &ctx->d_scores_out, (size_t)ctx->R * 4))) return rc;
    if (!ctx->has_list_model && !ctx->has_scalar_model) return fail(ctx, SF_ERR_INVALID, "no planning variable configured");
    if (ctx->has_list_model) {
        size_t lds = ((size_t)ctx->lm.dim + 31) / 32 * 4 + 16;
        hipLaunchKernelGGL(k_list_evaluate_all, dim3(ctx->R), dim3(256), lds, ctx->stream, ctx->lm,
                           ctx->d_scores_out, commit, d_parts);
    }
    if (ctx->has_scalar_model)  // mixed model: the scalar class adds its constraints to the list class's scores
        hipLaunchKernelGGL(k_scalar_evaluate_all, dim3(ctx->R), dim3(256), scalar_table_bytes(ctx), ctx->stream, ctx->sm,
                           ctx->d_scores_out, commit, ctx->has_list_model ? 1 : 0, d_parts);
    if (ctx->xown_level >= 0)  // the join of the two planning classes adds its level
        hipLaunchKernelGGL(k_cross_owner_evaluate_all, dim3(ctx->R), dim3(256), 0, ctx->stream, ctx->lm, ctx->sm.vals, ctx->sm.n, ctx->xown_level, ctx->xown_weight,
                           ctx->d_scores_out, commit, d_parts);
    if (ctx->has_list_model && ctx->pm.on)  // after the other constraints wrote their sums: adds its two levels
        hipLaunchKernelGGL(k_prec_evaluate_all, dim3(ctx->R), dim3(64), 0, ctx->stream, ctx->lm, ctx->pm, ctx->d_scores_out, commit, d_parts);
    HIPCHK(ctx, hipGetLastError());
    if (out) {
        std::vector<int64_t> tmp((size_t)ctx->R * ctx->levels);
        HIPCHK(ctx, hipMemcpyAsync(tmp.data(), ctx->d_scores_out, tmp.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::memcpy(out, tmp.data(), tmp.size() * 8);
    } else
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

extern "C" {

int32_t sf_initialize(sf_ctx* ctx, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (!ctx) return SF_ERR_INVALID;
    if (!ctx->initialized) {
        int n_list = 0, n_scalar = 0, ld = -1, sd = -1;
        for (auto& kv : ctx->classes) {
            if (kv.second.has_list) {
                ++n_list;
                ld = kv.first;
            }
            if (kv.second.has_scalar) {
                ++n_scalar;
                sd = kv.first;
            }
        }
        if (n_list + n_scalar == 0) return fail(ctx, SF_ERR_INVALID, "no planning variable configured");
        if (n_list > 1 || n_scalar > 1)
            return fail(ctx, SF_ERR_UNSUPPORTED, "at most one list class and one scalar class per context");
        int rc;
        if (n_list && (rc = build_list_model(ctx, ld))) return rc;
        if (n_scalar && (rc = build_scalar_model(ctx, sd))) return rc;
        if (n_list && n_scalar) {  // mixed model: one committed / best score, kept with the list class
            ctx->sm.score = ctx->lm.score;
            ctx->sm.best_score = ctx->lm.best_score;
        }
        for (auto& cs : ctx->constraints) {
            if (cs.kind != SF_C_CROSS_OWNER_MATCH) continue;
            if (!n_list || !n_scalar || cs.desc != sd || (int)cs.param != ld)
                return fail(ctx, SF_ERR_INVALID, "owner match: a join of the scalar class (descriptor_index) with the list class (param) of a mixed model");
            if (ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "one join of the two planning classes per context");
            if (ctx->sm.n_values != ctx->lm.V) return fail(ctx, SF_ERR_INVALID, "owner match: the scalar variable's values must be the list owner indices (n_values == owners)");
            if (ctx->lm.dim > ctx->sm.n) return fail(ctx, SF_ERR_INVALID, "owner match: the list elements must be entity ids of the scalar class");
            if (ctx->sm.n > 65534) return fail(ctx, SF_ERR_UNSUPPORTED, "owner match: more than 65534 entities");
            ctx->xown_level = cs.level;
            ctx->xown_weight = cs.weight;
            if ((rc = dalloc(ctx, &ctx->d_xown_tab, (size_t)ctx->R * ctx->sm.n))) return rc;
        }
        ctx->initialized = true;
    }
    return run_evaluate_all(ctx, out_scores, 1);
}

int32_t sf_evaluate_all(sf_ctx* ctx, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    return run_evaluate_all(ctx, out_scores, 0);
}

// ConstraintSet::evaluate_each (crates/solverforge-scoring/src/api/constraint_set/incremental.rs:172,237-244): one full
// recomputation, reported per declared constraint
int32_t sf_evaluate_each(sf_ctx* ctx, int32_t replica, int64_t* out_scores, int64_t* out_match_counts) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || replica < 0 || replica >= ctx->R || !out_scores || !out_match_counts)
        return fail(ctx, SF_ERR_INVALID, "bad sf_evaluate_each arguments");
    int rc;
    if (!ctx->d_each && (rc = dalloc(ctx, &ctx->d_each, (size_t)ctx->R * SF_EACH_WORDS))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_each, 0, (size_t)ctx->R * SF_EACH_WORDS * 8, ctx->stream));
    if ((rc = run_evaluate_all(ctx, nullptr, 0, ctx->d_each))) return rc;
    int64_t q[SF_EACH_WORDS];
    HIPCHK(ctx, hipMemcpy(q, ctx->d_each + (size_t)replica * SF_EACH_WORDS, sizeof(q), hipMemcpyDeviceToHost));
    // a class with uni programs: its value-cost components were folded into one device matrix; each keeps its own row here, from the host copies
    std::vector<int32_t> uni_vals;
    if (!ctx->uni_components.empty()) {
        uni_vals.resize((size_t)ctx->sm.n);
        HIPCHK(ctx, hipMemcpy(uni_vals.data(), ctx->sm.vals + (size_t)replica * ctx->sm.n, uni_vals.size() * 4, hipMemcpyDeviceToHost));
    }
    auto uni_row = [&](size_t constraint_index, int64_t& raw, int64_t& count) -> bool {
        for (auto& uc : ctx->uni_components) {
            if (uc.constraint_index != constraint_index) continue;
            raw = 0, count = 0;
            const size_t nv = (size_t)ctx->sm.n_values;
            for (size_t a = 0; a < uni_vals.size(); ++a)
                if (uni_vals[a] >= 0) raw += uc.cost[a * nv + (size_t)uni_vals[a]], count += uc.pass[a * nv + (size_t)uni_vals[a]];
            return true;
        }
        return false;
    };
    size_t i = 0;
    for (auto& cs : ctx->constraints) {
        int64_t raw = 0, count = 0;
        const bool on_list = ctx->has_list_model && cs.desc == ctx->list_desc;
        if (uni_row(i, raw, count)) {  // (scale / weight are inside `raw`)
            for (int k = 0; k < ctx->levels; ++k) out_scores[i * ctx->levels + k] = 0;
            out_scores[i * ctx->levels + cs.level] = (int64_t)(0 - (uint64_t)raw);
            out_match_counts[i] = count;
            ++i;
            continue;
        }
        switch (cs.kind) {
            case SF_C_ROUTE_CAPACITY: raw = q[0], count = ctx->lm.V; break;  // filter = every route (uni on the owners)
            case SF_C_ROUTE_DISTANCE: raw = q[1], count = ctx->lm.V; break;
            case SF_C_NOT_EXISTS_FLATTENED: raw = q[2], count = q[2]; break;
            case SF_C_UNI_UNASSIGNED: raw = q[3], count = q[16]; break;  // weighted sum / entities passing the filter
            case SF_C_CROSS_ADJACENT_EQUAL:
            case SF_C_CROSS_GROUP_EQUAL:
            case SF_C_PAIR_JOIN_:
            case SF_C_CROSS_QUEENS: raw = q[4], count = q[4]; break;
            case SF_C_CROSS_OWNER_MATCH: raw = q[17], count = q[17]; break;
            case SF_C_SELFJOIN_VALUE_EQUAL: raw = q[5], count = q[5]; break;
            case SF_C_GROUPED_VALUE_SUM:
            case SF_C_COMPLEMENTED_VALUE_SUM:
            case SF_C_LOAD_BALANCE_VALUE: raw = q[6], count = q[7]; break;
            case SF_C_BALANCE_VALUE: raw = q[6], count = q[7]; break;
            case SF_C_VALUE_COST: raw = q[8], count = q[9]; break;
            case SF_C_EXISTS_VALUE: raw = q[10], count = q[11]; break;
            case SF_C_RUNS_VALUE:
            case SF_C_PRESENCE_VALUE: raw = q[14], count = q[15]; break;
            case SF_C_LIST_PRECEDENCE_MAKESPAN: raw = q[12], count = q[12] + (q[13] > 0 ? 1 : 0); break;  // match_count_from_state (:96-99)
            default: return fail(ctx, SF_ERR_UNSUPPORTED, "constraint kind in sf_evaluate_each");
        }
        (void)on_list;
        for (int k = 0; k < ctx->levels; ++k) out_scores[i * ctx->levels + k] = 0;
        // penalties; the balance constraint's base score is already inside `raw` (round(base * standard deviation))
        out_scores[i * ctx->levels + cs.level] = cs.kind == SF_C_BALANCE_VALUE ? (int64_t)(0 - (uint64_t)raw) : (int64_t)(0 - (uint64_t)cs.weight * (uint64_t)raw);
        if (cs.kind == SF_C_LIST_PRECEDENCE_MAKESPAN) {  // two levels: -hard penalty, -makespan (`weight` holds the makespan level)
            out_scores[i * ctx->levels + cs.level] = -q[12];
            out_scores[i * ctx->levels + (int)cs.weight] = -q[13];
        }
        out_match_counts[i] = count;
        ++i;
    }
    return SF_OK;
}

int32_t sf_get_scores(sf_ctx* ctx, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || !out_scores) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    return download_scores(ctx, ctx->has_list_model ? ctx->lm.score : ctx->sm.score, out_scores);
}

// SF_MOVE_LIST_RUIN entries of a host batch: one wavefront each (csrc/sf_construct.hip)
static hipError_t launch_ruin_moves(sf_ctx* ctx, int replica, const int32_t* d_moves, const std::vector<int32_t>& which, int64_t* d_sc, int32_t* d_do,
                                    int commit) {
    const SelectorSpec* rs = ruin_selector(ctx);
    const int skip_empty = rs ? rs->skip_empty : 0;
    int32_t* d_idx = nullptr;
    hipError_t e = hipMalloc((void**)&d_idx, which.size() * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, which.data(), which.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    const RuinMoveCarve cv(ctx->lm.V, ctx->lm.n_cap);
    if (e == hipSuccess) {
        if (ctx->levels <= 2) {
            auto kern = k_list_ruin_moves<2>;
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
            if (e == hipSuccess)
                hipLaunchKernelGGL(kern, dim3((unsigned)which.size()), dim3(64), cv.total, ctx->stream, ctx->lm, replica, d_moves, d_idx, d_sc, d_do, skip_empty, commit);
        } else {
            auto kern = k_list_ruin_moves<4>;
            e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
            if (e == hipSuccess)
                hipLaunchKernelGGL(kern, dim3((unsigned)which.size()), dim3(64), cv.total, ctx->stream, ctx->lm, replica, d_moves, d_idx, d_sc, d_do, skip_empty, commit);
        }
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_idx);
    return e;
}

// SF_MOVE_LIST_RUIN records on a precedence model: k_prec_ruin_moves, at most R records per launch (one scratch slot each)
static int ensure_plf(sf_ctx* ctx);
static int launch_prec_ruin_moves(sf_ctx* ctx, int replica, const int32_t* d_moves, const std::vector<int32_t>& which, int64_t* d_sc, int32_t* d_do, int commit) {
    if (ctx->lm.dist_level >= 0 || ctx->lm.cap_level >= 0)
        return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin move on a precedence model with distance / capacity constraints");
    if (int rc = ensure_plf(ctx)) return rc;
    const size_t lds = (((size_t)ctx->lm.V + 1 + 3) & ~(size_t)3) * 4 + (size_t)ctx->lm.n_cap * 2 + 16;
    if (ctx->lm.n_cap > 65535 || lds > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin moves: the list class must fit one wave's LDS slice with 16-bit elements");
    const SelectorSpec* rs = ruin_selector(ctx);
    const int skip_empty = rs ? rs->skip_empty : 0;
    const int lvl_order = ctx->pm.hard_level < ctx->pm.mk_level ? 0 : (ctx->pm.hard_level > ctx->pm.mk_level ? 1 : 2);
    int32_t* d_idx = nullptr;
    hipError_t e = hipMalloc((void**)&d_idx, which.size() * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_idx, which.data(), which.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_prec_ruin_moves, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (size_t base = 0; base < which.size() && e == hipSuccess; base += (size_t)ctx->R) {
        const unsigned chunk = (unsigned)std::min<size_t>((size_t)ctx->R, which.size() - base);
        hipLaunchKernelGGL(k_prec_ruin_moves, dim3(chunk), dim3(64), lds, ctx->stream, ctx->lm, ctx->pm, ctx->plf, replica, d_moves, d_idx + base, d_sc, d_do, commit,
                           lvl_order, ctx->prec_policy ? 1 : 0, skip_empty);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_idx);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return SF_OK;
}

int32_t sf_step_evaluate(sf_ctx* ctx, int32_t replica, const sf_move_t* moves, int64_t n, int64_t* out_scores,
                         int32_t* out_doable) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (replica < 0 || replica >= ctx->R || n < 0 || !moves || !out_scores || !out_doable)
        return fail(ctx, SF_ERR_INVALID, "bad sf_step_evaluate arguments");
    if (ctx->xown_level >= 0)
        for (int64_t i = 0; i < n; ++i)
            if (moves[i].kind == SF_MOVE_LIST_RUIN) return fail(ctx, SF_ERR_UNSUPPORTED, "the join of the two planning classes is not priced by a ruin's recreate");
    if (n == 0) return SF_OK;
    // one allocation per call, released on every path (hipFree(nullptr) is a no-op)
    int32_t* d_moves = nullptr;
    int64_t* d_sc = nullptr;
    int32_t* d_do = nullptr;
    auto release = [&]() {
        (void)hipFree(d_moves);
        (void)hipFree(d_sc);
        (void)hipFree(d_do);
    };
    hipError_t ea = hipMalloc((void**)&d_moves, (size_t)n * 24);
    if (ea == hipSuccess) ea = hipMalloc((void**)&d_sc, (size_t)n * ctx->levels * 8);
    if (ea == hipSuccess) ea = hipMalloc((void**)&d_do, (size_t)n * 4);
    if (ea == hipSuccess) ea = hipMemcpyAsync(d_moves, moves, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream);
    if (ea != hipSuccess) {
        release();
        return fail(ctx, SF_ERR_HIP, hipGetErrorString(ea));
    }
    int grid = (int)((n + 255) / 256);
    const int mixed = ctx->has_list_model && ctx->has_scalar_model;
    if (mixed) {
        (void)hipMemsetAsync(d_sc, 0, (size_t)n * ctx->levels * 8, ctx->stream);
        (void)hipMemsetAsync(d_do, 0, (size_t)n * 4, ctx->stream);
    }
    if (ctx->has_list_model)
        hipLaunchKernelGGL(k_list_evaluate_moves, dim3(grid), dim3(256), 0, ctx->stream, ctx->lm, replica, d_moves, n, d_sc, d_do, mixed);
    if (ctx->has_scalar_model)
        hipLaunchKernelGGL(k_scalar_evaluate_moves, dim3(grid), dim3(256), scalar_table_bytes(ctx), ctx->stream, ctx->sm, replica, d_moves, n, d_sc, d_do, mixed);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && ctx->has_list_model) {  // list ruin moves: scored by their own kernel, one wavefront per move
        std::vector<int32_t> which;
        for (int64_t i = 0; i < n; ++i)
            if (moves[i].kind == SF_MOVE_LIST_RUIN) which.push_back((int32_t)i);
        if (!which.empty() && ctx->pm.on) {  // precedence model: the recreate is scored by the precedence constraint (k_prec_ruin_moves)
            const int rc2 = launch_prec_ruin_moves(ctx, replica, d_moves, which, d_sc, d_do, 0);
            if (rc2) {
                release();
                return rc2;
            }
        } else if (!which.empty()) {
            if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536 || RuinMoveCarve(ctx->lm.V, ctx->lm.n_cap).total > SF_LDS_BUDGET) {
                release();
                return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin moves: the list class must fit one wave's LDS slice with 16-bit elements");
            }
            e = launch_ruin_moves(ctx, replica, d_moves, which, d_sc, d_do, 0);
        }
    }
    if (e == hipSuccess && ctx->has_list_model && ctx->pm.on) {  // precedence delta of every doable list move: one wavefront per record
        const PrecMoveCarve cv(ctx->lm.V, ctx->lm.n_cap);
        if (ctx->lm.n_cap > 65535 || cv.total > SF_LDS_BUDGET) {
            release();
            return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence moves: the list class must fit one wave's LDS slice with 16-bit elements");
        }
        e = hipFuncSetAttribute((const void*)k_prec_evaluate_moves, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
        for (int64_t base = 0; base < n && e == hipSuccess; base += ctx->R) {
            const int chunk = (int)std::min<int64_t>(ctx->R, n - base);
            hipLaunchKernelGGL(k_prec_evaluate_moves, dim3(chunk), dim3(64), cv.total, ctx->stream, ctx->lm, ctx->pm, replica, d_moves, base, d_sc, d_do);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess && ctx->xown_level >= 0) {  // the join of the two planning classes: its delta from the move's coordinates (k_cross_owner_evaluate_moves)
        hipLaunchKernelGGL(k_cross_owner_holders, dim3(1), dim3(256), 0, ctx->stream, ctx->lm, replica, ctx->sm.n, ctx->d_xown_tab);
        hipLaunchKernelGGL(k_cross_owner_evaluate_moves, dim3(grid), dim3(256), 0, ctx->stream, ctx->lm, ctx->sm.vals, ctx->sm.n, ctx->d_xown_tab, replica, d_moves, n,
                           ctx->xown_level, ctx->xown_weight, d_sc, d_do);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_sc, (size_t)n * ctx->levels * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_doable, d_do, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release();
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return SF_OK;
}

static int xown_price(sf_ctx* ctx, int32_t replica, const sf_move_t* records, int64_t n_records, const int64_t* compound_offsets);
static void xown_commit(sf_ctx* ctx, int32_t replica);
// ScalarCandidateProvider surface: multi-edit candidates scored as ONE CompoundScalarMove each
int32_t sf_step_evaluate_compound(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, int64_t n,
                                  int64_t* out_scores, int32_t* out_doable) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->has_scalar_model) return fail(ctx, SF_ERR_INVALID, "compound scalar candidates need a scalar variable");
    if (replica < 0 || replica >= ctx->R || n < 0 || !offsets || !out_scores || !out_doable)
        return fail(ctx, SF_ERR_INVALID, "bad sf_step_evaluate_compound arguments");
    if (ctx->sm.grp_level >= 0 && ctx->sm.grp_mode >= 1)
        return fail(ctx, SF_ERR_UNSUPPORTED, "compound candidates on a load_balance / balance model (floating-point aggregate) are not chained on the device");
    if (ctx->sm.run_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "compound candidates on a consecutive-runs model are not chained on the device");
    if (n == 0) return SF_OK;
    if (offsets[0] != 0) return fail(ctx, SF_ERR_INVALID, "offsets[0] must be 0");
    for (int64_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SF_ERR_INVALID, "offsets must not decrease");
        if (offsets[i + 1] - offsets[i] > SF_COMPOUND_MAX) return fail(ctx, SF_ERR_UNSUPPORTED, "at most 8 edits per compound candidate on the device");
    }
    const int64_t total = offsets[n];
    if (total > 0 && !edits) return fail(ctx, SF_ERR_INVALID, "edits is NULL");
    for (int64_t k = 0; k < total; ++k)
        if (edits[k].kind != SF_MOVE_CHANGE) return fail(ctx, SF_ERR_INVALID, "a ScalarEdit is a SF_MOVE_CHANGE-shaped record");
    int32_t* d_edits = nullptr;
    int64_t *d_off = nullptr, *d_sc = nullptr;
    int32_t* d_do = nullptr;
    auto release = [&]() {
        (void)hipFree(d_edits);
        (void)hipFree(d_off);
        (void)hipFree(d_sc);
        (void)hipFree(d_do);
    };
    hipError_t ea = hipMalloc((void**)&d_edits, (size_t)(total > 0 ? total : 1) * 24);
    if (ea == hipSuccess) ea = hipMalloc((void**)&d_off, (size_t)(n + 1) * 8);
    if (ea == hipSuccess) ea = hipMalloc((void**)&d_sc, (size_t)n * ctx->levels * 8);
    if (ea == hipSuccess) ea = hipMalloc((void**)&d_do, (size_t)n * 4);
    if (ea == hipSuccess && total > 0) ea = hipMemcpyAsync(d_edits, edits, (size_t)total * 24, hipMemcpyHostToDevice, ctx->stream);
    if (ea == hipSuccess) ea = hipMemcpyAsync(d_off, offsets, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (ea != hipSuccess) {
        release();
        return fail(ctx, SF_ERR_HIP, hipGetErrorString(ea));
    }
    hipLaunchKernelGGL(k_scalar_evaluate_compound, dim3((int)((n + 255) / 256)), dim3(256), scalar_table_bytes(ctx), ctx->stream, ctx->sm, replica,
                       d_edits, d_off, n, d_sc, d_do);
    if (ctx->xown_level >= 0) {  // the join of the two planning classes: a scalar edit changes the A side's key
        hipLaunchKernelGGL(k_cross_owner_holders, dim3(1), dim3(256), 0, ctx->stream, ctx->lm, replica, ctx->sm.n, ctx->d_xown_tab);
        hipLaunchKernelGGL(k_cross_owner_evaluate_compound, dim3((int)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->sm.vals, ctx->sm.n, ctx->d_xown_tab, replica, d_edits,
                           d_off, n, ctx->levels, ctx->xown_level, ctx->xown_weight, d_sc, d_do);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_sc, (size_t)n * ctx->levels * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_doable, d_do, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release();
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return SF_OK;
}

int32_t sf_apply_compound(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, int64_t n_edits) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || !edits || replica < 0 || replica >= ctx->R) return fail(ctx, SF_ERR_INVALID, "bad sf_apply_compound arguments");
    if (!ctx->has_scalar_model) return fail(ctx, SF_ERR_INVALID, "compound scalar candidates need a scalar variable");
    if (n_edits <= 0) return fail(ctx, SF_ERR_INVALID, "move is not doable");
    if (n_edits > SF_COMPOUND_MAX) return fail(ctx, SF_ERR_UNSUPPORTED, "at most 8 edits per compound candidate on the device");
    if (ctx->sm.grp_level >= 0 && ctx->sm.grp_mode >= 1)
        return fail(ctx, SF_ERR_UNSUPPORTED, "compound candidates on a load_balance / balance model (floating-point aggregate) are not chained on the device");
    if (ctx->sm.run_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "compound candidates on a consecutive-runs model are not chained on the device");
    for (int64_t k = 0; k < n_edits; ++k)
        if (edits[k].kind != SF_MOVE_CHANGE) return fail(ctx, SF_ERR_INVALID, "a ScalarEdit is a SF_MOVE_CHANGE-shaped record");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    int32_t* d_edits = nullptr;
    hipError_t ea = hipMalloc((void**)&d_edits, (size_t)n_edits * 24);
    if (ea == hipSuccess) ea = hipMemcpyAsync(d_edits, edits, (size_t)n_edits * 24, hipMemcpyHostToDevice, ctx->stream);
    if (ea != hipSuccess) {
        (void)hipFree(d_edits);
        return fail(ctx, SF_ERR_HIP, hipGetErrorString(ea));
    }
    {
        const int64_t one_candidate[2] = {0, n_edits};
        if ((rc = xown_price(ctx, replica, edits, n_edits, one_candidate))) {
            (void)hipFree(d_edits);
            return rc;
        }
    }
    hipLaunchKernelGGL(k_scalar_apply_compound, dim3(1), dim3(64), scalar_table_bytes(ctx), ctx->stream, ctx->sm, replica, d_edits, (int)n_edits,
                       ctx->d_ok);
    xown_commit(ctx, replica);
    int32_t ok = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&ok, ctx->d_ok, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_edits);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    if (!ok) return fail(ctx, SF_ERR_INVALID, "move is not doable");
    return SF_OK;
}

// One host-driven local-search step over a ScalarCandidateProvider's output (GroupedScalarMoveSelector; see the header).
int32_t sf_step_decide(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, int64_t n, int32_t group_name_len,
                       int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_n_kept, int64_t* out_scores, int32_t* out_flags,
                       int64_t* out_consumed, int64_t* out_selected) {
    return sf_step_decide_gated(ctx, replica, edits, offsets, nullptr, n, group_name_len, max_moves_per_step, out_kept, out_n_kept, out_scores, out_flags,
                                out_consumed, out_selected);
}
static int32_t step_decide_impl(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n,
                                int32_t group_name_len, int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_n_kept, int64_t* out_scores,
                                int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected, bool cursor_order);
// the same step with Move::requires_hard_improvement / requires_score_improvement per candidate (gates[i]: bit 0 / bit 1)
int32_t sf_step_decide_gated(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n,
                             int32_t group_name_len, int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_n_kept, int64_t* out_scores,
                             int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected) {
    return step_decide_impl(ctx, replica, edits, offsets, gates, n, group_name_len, max_moves_per_step, out_kept, out_n_kept, out_scores, out_flags, out_consumed,
                            out_selected, false);
}
// the step over a cursor's own pull order (RuntimeProviderCursor, runtime/provider_cursor.rs:447-466): no activation is restated here --
// the cursor has rotated, normalised, deduplicated per provider scope and capped its store, and pushed doable moves only
// (provider_cursor.rs:420-437) -- so candidate i is pull i; see the header
int32_t sf_step_decide_cursor(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n,
                              int64_t* out_scores, int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected) {
    std::vector<int64_t> kept((size_t)(n > 0 ? n : 1));
    int64_t nk = 0;
    return step_decide_impl(ctx, replica, edits, offsets, gates, n, 0, 0, kept.data(), &nk, out_scores, out_flags, out_consumed, out_selected, true);
}
static int32_t step_decide_impl(sf_ctx* ctx, int32_t replica, const sf_move_t* edits, const int64_t* offsets, const int32_t* gates, int64_t n,
                                int32_t group_name_len, int64_t max_moves_per_step, int64_t* out_kept, int64_t* out_n_kept, int64_t* out_scores,
                                int32_t* out_flags, int64_t* out_consumed, int64_t* out_selected, bool cursor_order) {
    DeviceGuard _dev(ctx);
    if (ctx && ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_step_decide_gated: a model with the join of its two planning classes is searched by the fused engine only");
    if (!ctx || !ctx->initialized || replica < 0 || replica >= ctx->R || n < 0 || !offsets || !out_kept || !out_n_kept || !out_scores || !out_flags ||
        !out_consumed || !out_selected)
        return fail(ctx, SF_ERR_INVALID, "bad sf_step_decide arguments");
    if (!ctx->has_scalar_model || ctx->has_list_model) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_step_decide: scalar-only models (ScalarCandidate edits)");
    if (ctx->sm.grp_level >= 0 && ctx->sm.grp_mode >= 1)
        return fail(ctx, SF_ERR_UNSUPPORTED, "compound candidates on a load_balance / balance model (floating-point aggregate) are not chained on the device");
    if (ctx->sm.run_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "compound candidates on a consecutive-runs model are not chained on the device");
    if (ctx->cfg.acceptor != SF_ACCEPT_HILL_CLIMBING && ctx->cfg.acceptor != SF_ACCEPT_LATE_ACCEPTANCE && ctx->cfg.acceptor != SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE)
        return fail(ctx, SF_ERR_UNSUPPORTED, "sf_step_decide: HillClimbing, LateAcceptance or DiversifiedLateAcceptance");
    if (offsets[0] != 0) return fail(ctx, SF_ERR_INVALID, "offsets[0] must be 0");
    if (n >= ((int64_t)1 << 31)) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_step_decide: fewer than 2^31 candidates");
    for (int64_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(ctx, SF_ERR_INVALID, "offsets must not decrease");
    if (n > 0 && offsets[n] > 0 && !edits) return fail(ctx, SF_ERR_INVALID, "edits is NULL");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    SearchParams p = ctx->sp;
    fill_search_params(ctx, p);
    const ClassSpec& c = ctx->classes[ctx->scalar_desc];
    const int ne = ctx->sm.n;
    // the step's MoveStreamContext and the replica's working values (the cursor filters by is_doable_on)
    std::vector<int32_t> vals((size_t)ne);
    uint64_t step_index = 0, draws = 0;
    HIPCHK(ctx, hipMemcpyAsync(vals.data(), ctx->sm.vals + (size_t)replica * ne, (size_t)ne * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&step_index, p.step_index + replica, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(&draws, p.seed_draws + replica, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t sseed = step_seed(p.random_seed + (uint64_t)replica, draws);
    if (ctx->d_explicit && (int64_t)draws < ctx->n_explicit) {
        HIPCHK(ctx, hipMemcpy(&sseed, ctx->d_explicit + (size_t)replica * ctx->n_explicit + draws, 8, hipMemcpyDeviceToHost));
    }
    const StreamCtx sctx{step_index, sseed, p.order};
    const int64_t cap = max_moves_per_step > 0 ? max_moves_per_step : 256;  // candidate-backed group (grouped_scalar.rs:27-40)
    // GroupedScalarCursor::activate, Candidates arm (grouped_scalar.rs:122-176)
    std::vector<int64_t> kept;
    auto legal = [&](int32_t e, int32_t to) {
        if (e < 0 || e >= ne) return false;
        if (to < 0) return to == -1 && c.allows_unassigned != 0;
        if (to >= c.n_values) return false;
        if (c.value_off.empty()) return true;
        for (uint32_t q = c.value_off[(size_t)e]; q < c.value_off[(size_t)e + 1]; ++q)
            if (c.value_list[q] == to) return true;
        return false;
    };
    for (int64_t o = 0; cursor_order && o < n; ++o) {  // a cursor's store: pull order, nothing skipped; malformed records are the caller's error
        const int64_t b = offsets[o], e = offsets[o + 1];
        if (e == b) return fail(ctx, SF_ERR_INVALID, "sf_step_decide_cursor: a candidate without edits (the cursor normalises its store)");
        if (e - b > SF_COMPOUND_MAX) return fail(ctx, SF_ERR_UNSUPPORTED, "at most 8 edits per compound candidate on the device");
        for (int64_t k = b; k < e; ++k) {
            if (edits[k].kind != SF_MOVE_CHANGE) return fail(ctx, SF_ERR_INVALID, "a ScalarEdit is a SF_MOVE_CHANGE-shaped record");
            for (int64_t j = b; j < k; ++j)
                if (edits[j].a == edits[k].a) return fail(ctx, SF_ERR_INVALID, "sf_step_decide_cursor: two edits on one entity (the cursor normalises its store)");
            if (!legal(edits[k].a, edits[k].value)) return fail(ctx, SF_ERR_INVALID, "sf_step_decide_cursor: an edit outside the entity's value range");
        }
        kept.push_back(o);
    }
    for (int64_t o = 0; !cursor_order && o < n && (int64_t)kept.size() < cap; ++o) {
        const int64_t idx = (int64_t)sctx.selection_index((uint32_t)o, (uint32_t)n, 0xC0A1E5CEAAA00001ULL ^ (uint64_t)group_name_len);  // apply_selection_order
        const int64_t b = offsets[idx], e = offsets[idx + 1];
        if (e == b) continue;
        if (e - b > SF_COMPOUND_MAX) return fail(ctx, SF_ERR_UNSUPPORTED, "at most 8 edits per compound candidate on the device");
        bool ok = true, changes = false;
        for (int64_t k = b; k < e && ok; ++k) {
            if (edits[k].kind != SF_MOVE_CHANGE) return fail(ctx, SF_ERR_INVALID, "a ScalarEdit is a SF_MOVE_CHANGE-shaped record");
            for (int64_t j = b; j < k; ++j) ok = ok && edits[j].a != edits[k].a;  // two edits on one (descriptor, entity, variable)
            ok = ok && legal(edits[k].a, edits[k].value);
            if (ok) changes = changes || vals[(size_t)edits[k].a] != edits[k].value;
        }
        if (!ok || !changes) continue;
        bool seen = false;
        for (int64_t q : kept) {
            if (offsets[q + 1] - offsets[q] != e - b) continue;
            bool same = true;
            for (int64_t k = 0; k < e - b && same; ++k) same = edits[offsets[q] + k].a == edits[b + k].a && edits[offsets[q] + k].value == edits[b + k].value;
            seen = seen || same;
        }
        if (seen) continue;
        kept.push_back(idx);
    }
    const int64_t nk = (int64_t)kept.size();
    *out_n_kept = nk;
    for (int64_t i = 0; i < nk; ++i) out_kept[i] = kept[(size_t)i];
    // kept candidates as their own CSR
    std::vector<sf_move_t> kedits;
    std::vector<int64_t> koff(1, 0);
    for (int64_t q : kept) {
        for (int64_t k = offsets[q]; k < offsets[q + 1]; ++k) kedits.push_back(edits[k]);
        koff.push_back((int64_t)kedits.size());
    }
    int32_t* d_edits = nullptr;
    int64_t *d_off = nullptr, *d_sc = nullptr, *d_res = nullptr;
    int32_t *d_do = nullptr, *d_fl = nullptr, *d_gates = nullptr;
    auto release = [&]() {
        (void)hipFree(d_edits), (void)hipFree(d_off), (void)hipFree(d_sc), (void)hipFree(d_res), (void)hipFree(d_do), (void)hipFree(d_fl), (void)hipFree(d_gates);
    };
    std::vector<int32_t> kgates;  // in pull order
    if (gates)
        for (int64_t q : kept) kgates.push_back(gates[q]);
    const size_t nk1 = (size_t)(nk > 0 ? nk : 1);
    hipError_t e = hipMalloc((void**)&d_edits, (kedits.empty() ? 1 : kedits.size()) * 24);
    if (e == hipSuccess) e = hipMalloc((void**)&d_off, (size_t)(nk + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_sc, nk1 * ctx->levels * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&d_do, nk1 * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_fl, nk1 * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_res, 16);
    if (e == hipSuccess && !kedits.empty()) e = hipMemcpyAsync(d_edits, kedits.data(), kedits.size() * 24, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, koff.data(), (size_t)(nk + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_fl, 0, nk1 * 4, ctx->stream);
    if (e == hipSuccess && !kgates.empty()) {
        e = hipMalloc((void**)&d_gates, kgates.size() * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_gates, kgates.data(), kgates.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e != hipSuccess) {
        release();
        return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    }
    if (nk > 0)
        hipLaunchKernelGGL(k_scalar_evaluate_compound, dim3((int)((nk + 255) / 256)), dim3(256), scalar_table_bytes(ctx), ctx->stream, ctx->sm, replica, d_edits,
                           d_off, nk, d_sc, d_do);
    hipLaunchKernelGGL(k_scalar_step_decide, dim3(1), dim3(64), scalar_table_bytes(ctx), ctx->stream, ctx->sm, p, replica, d_edits, d_off, nk, d_sc, d_do, d_fl,
                       d_res, (const int32_t*)d_gates, ctx->hard_levels);
    e = hipGetLastError();
    int64_t res[2] = {0, -1};
    if (e == hipSuccess) e = hipMemcpyAsync(res, d_res, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && nk > 0) e = hipMemcpyAsync(out_scores, d_sc, (size_t)nk * ctx->levels * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && nk > 0) e = hipMemcpyAsync(out_flags, d_fl, (size_t)nk * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release();
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    *out_consumed = res[0];
    *out_selected = res[1];
    return SF_OK;
}

// The join of the two planning classes under sf_apply / sf_apply_compound: its delta is priced on the state BEFORE the move (into ctx->d_xown_delta,
// SF_MAX_LEVELS words), and added to the committed score by xown_commit once the apply kernel has said the move went through (ctx->d_ok).
static int xown_price(sf_ctx* ctx, int32_t replica, const sf_move_t* records, int64_t n_records, const int64_t* compound_offsets) {
    if (ctx->xown_level < 0) return SF_OK;
    if (!ctx->d_xown_delta) {
        int rc = dalloc(ctx, &ctx->d_xown_delta, (size_t)SF_MAX_LEVELS);
        if (rc) return rc;
    }
    int32_t* d_rec = nullptr;
    int64_t* d_off = nullptr;
    hipError_t e = hipMalloc((void**)&d_rec, (size_t)n_records * 24);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rec, records, (size_t)n_records * 24, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(ctx->d_xown_delta, 0, (size_t)SF_MAX_LEVELS * 8, ctx->stream);
    if (e == hipSuccess && compound_offsets) {
        e = hipMalloc((void**)&d_off, 16);
        if (e == hipSuccess) e = hipMemcpyAsync(d_off, compound_offsets, 16, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_cross_owner_holders, dim3(1), dim3(256), 0, ctx->stream, ctx->lm, replica, ctx->sm.n, ctx->d_xown_tab);
        if (compound_offsets)  // ONE compound candidate: records [0, n_records)
            hipLaunchKernelGGL(k_cross_owner_evaluate_compound, dim3(1), dim3(256), 0, ctx->stream, ctx->sm.vals, ctx->sm.n, ctx->d_xown_tab, replica, d_rec, d_off,
                               (int64_t)1, ctx->levels, ctx->xown_level, ctx->xown_weight, ctx->d_xown_delta, (const int32_t*)nullptr);
        else
            hipLaunchKernelGGL(k_cross_owner_evaluate_moves, dim3(1), dim3(256), 0, ctx->stream, ctx->lm, ctx->sm.vals, ctx->sm.n, ctx->d_xown_tab, replica, d_rec,
                               (int64_t)1, ctx->xown_level, ctx->xown_weight, ctx->d_xown_delta, (const int32_t*)nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (the host buffers are released below)
    (void)hipFree(d_rec);
    (void)hipFree(d_off);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return SF_OK;
}
static void xown_commit(sf_ctx* ctx, int32_t replica) {
    if (ctx->xown_level < 0) return;
    hipLaunchKernelGGL(k_cross_owner_commit, dim3(1), dim3(1), 0, ctx->stream, ctx->lm.score + (size_t)replica * 4 + ctx->xown_level,
                       ctx->d_xown_delta + ctx->xown_level, ctx->d_ok);
}

int32_t sf_apply(sf_ctx* ctx, int32_t replica, const sf_move_t* mv) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || !mv || replica < 0 || replica >= ctx->R)
        return fail(ctx, SF_ERR_INVALID, "bad sf_apply arguments");
    if (ctx->xown_level >= 0 && mv->kind == SF_MOVE_LIST_RUIN) return fail(ctx, SF_ERR_UNSUPPORTED, "the join of the two planning classes is not priced by a ruin's recreate");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    if (mv->kind == SF_MOVE_LIST_RUIN) {  // committed ruin + recreate: its own kernel (one wavefront)
        if (!ctx->has_list_model) return fail(ctx, SF_ERR_INVALID, "list move on a model without a list variable");
        if (ctx->pm.on) {  // the recreate by the precedence constraint; the committed scores are refreshed from the new lists
            int32_t* d_mv = nullptr;
            int64_t* d_sc = nullptr;
            int32_t* d_do = nullptr;
            hipError_t e = hipMalloc((void**)&d_mv, 24);
            if (e == hipSuccess) e = hipMalloc((void**)&d_sc, 4 * 8);
            if (e == hipSuccess) e = hipMalloc((void**)&d_do, 4);
            if (e == hipSuccess) e = hipMemcpyAsync(d_mv, mv, 24, hipMemcpyHostToDevice, ctx->stream);
            int32_t ok = 0;
            int rc2 = SF_OK;
            if (e == hipSuccess) rc2 = launch_prec_ruin_moves(ctx, replica, d_mv, std::vector<int32_t>{0}, d_sc, d_do, 1);
            if (e == hipSuccess && rc2 == SF_OK) e = hipMemcpy(&ok, d_do, 4, hipMemcpyDeviceToHost);
            (void)hipFree(d_mv);
            (void)hipFree(d_sc);
            (void)hipFree(d_do);
            if (rc2) return rc2;
            if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
            if (!ok) return fail(ctx, SF_ERR_INVALID, "move is not doable");
            return run_evaluate_all(ctx, nullptr, 1);
        }
        if (ctx->has_scalar_model) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_apply of a list ruin on a mixed model");
        if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536 || RuinMoveCarve(ctx->lm.V, ctx->lm.n_cap).total > SF_LDS_BUDGET)
            return fail(ctx, SF_ERR_UNSUPPORTED, "list ruin moves: the list class must fit one wave's LDS slice with 16-bit elements");
        int32_t* d_mv = nullptr;
        int64_t* d_sc = nullptr;
        int32_t* d_do = nullptr;
        hipError_t e = hipMalloc((void**)&d_mv, 24);
        if (e == hipSuccess) e = hipMalloc((void**)&d_sc, 4 * 8);
        if (e == hipSuccess) e = hipMalloc((void**)&d_do, 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_mv, mv, 24, hipMemcpyHostToDevice, ctx->stream);
        int32_t ok = 0;
        if (e == hipSuccess) e = launch_ruin_moves(ctx, replica, d_mv, std::vector<int32_t>{0}, d_sc, d_do, 1);
        if (e == hipSuccess) e = hipMemcpy(&ok, d_do, 4, hipMemcpyDeviceToHost);
        (void)hipFree(d_mv);
        (void)hipFree(d_sc);
        (void)hipFree(d_do);
        if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
        if (!ok) return fail(ctx, SF_ERR_INVALID, "move is not doable");
        return SF_OK;
    }
    if (mv->kind == SF_MOVE_LIST_MULTI_SWAP) {  // the lists are pairwise different: the swaps commute, so they are committed one after the other
        if (!ctx->has_list_model) return fail(ctx, SF_ERR_INVALID, "list move on a model without a list variable");
        if (mv->a < 1 || mv->a > 3) return fail(ctx, SF_ERR_INVALID, "multi-swap: 1..3 swaps");
        sf_move_t one[3];
        const int32_t words[3] = {mv->a_pos, mv->b, mv->b_pos};
        for (int q = 0; q < mv->a; ++q) {
            const uint32_t w = (uint32_t)words[q];
            const int32_t dl = (int32_t)(int8_t)(((uint32_t)mv->value >> (8 * q)) & 0xFFu);
            one[q] = sf_move_t{SF_MOVE_LIST_SWAP, (int32_t)(w & 0xFFFFu), (int32_t)(w >> 16), (int32_t)(w & 0xFFFFu), (int32_t)(w >> 16) + dl, -1};
            for (int q2 = 0; q2 < q; ++q2)
                if (one[q2].a == one[q].a) return fail(ctx, SF_ERR_INVALID, "multi-swap: the swaps must touch pairwise different lists");
            if (dl == 0 || one[q].b_pos < 0) return fail(ctx, SF_ERR_INVALID, "multi-swap: a swap needs two different positions");
        }
        for (int q = 0; q < mv->a; ++q) {
            const int32_t rc2 = sf_apply(ctx, replica, &one[q]);
            if (rc2 != SF_OK) {
                for (int q2 = q - 1; q2 >= 0; --q2) (void)sf_apply(ctx, replica, &one[q2]);  // a swap is its own inverse
                return rc2;
            }
        }
        return SF_OK;
    }
    const bool list_move = (mv->kind >= SF_MOVE_LIST_CHANGE && mv->kind <= SF_MOVE_KOPT) || mv->kind == SF_MOVE_LIST_PERMUTE;
    if (list_move && !ctx->has_list_model) return fail(ctx, SF_ERR_INVALID, "list move on a model without a list variable");
    if (!list_move && !ctx->has_scalar_model) return fail(ctx, SF_ERR_INVALID, "scalar move on a model without a scalar variable");
    if (list_move) {
        if (mv->a < 0 || mv->a >= ctx->lm.V || mv->b < 0 || (mv->kind != SF_MOVE_KOPT && mv->b >= ctx->lm.V) || mv->a_pos < 0 ||
            mv->b_pos < 0)
            return fail(ctx, SF_ERR_INVALID, "move out of range");
        if (mv->kind == SF_MOVE_KOPT && (mv->value < 0 || mv->value >= 7))
            return fail(ctx, SF_ERR_INVALID, "3-opt move: value is the reconnection pattern 0..6");
        if (mv->kind == SF_MOVE_LIST_PERMUTE && (mv->a != mv->b || mv->b_pos - mv->a_pos < 2 || mv->b_pos - mv->a_pos > 8 || mv->value < 1))
            return fail(ctx, SF_ERR_INVALID, "list permute move: a window of 2..8 positions of one list and a permutation rank >= 1");
        if (mv->kind == SF_MOVE_SUBLIST_CHANGE && (mv->value <= mv->a_pos || mv->value - mv->a_pos > 255))
            return fail(ctx, SF_ERR_INVALID, "sublist move: value must be the segment end (segment of 1..255 elements)");
        if (mv->kind == SF_MOVE_SUBLIST_SWAP && (mv->value <= 0 || (mv->value & 0xFFFF) == 0 || (mv->value & 0xFFFF) > 255 ||
                                                 (mv->value >> 16) == 0 || (mv->value >> 16) > 255))
            return fail(ctx, SF_ERR_INVALID, "sublist swap: value packs the two segment sizes (1..255 each)");
        if ((rc = xown_price(ctx, replica, mv, 1, nullptr))) return rc;
        hipLaunchKernelGGL(k_list_apply, dim3(1), dim3(256), 0, ctx->stream, ctx->lm, replica, mv->kind,
                           (uint32_t)mv->a, (uint32_t)mv->a_pos, (uint32_t)mv->b, (uint32_t)mv->b_pos,
                           (uint32_t)(mv->value > 0 ? mv->value : 0), ctx->d_ok);
        if (ctx->pm.on)  // a move that was not doable left the lists alone: the refresh then changes nothing
            hipLaunchKernelGGL(k_prec_after_apply, dim3(1), dim3(64), 0, ctx->stream, ctx->lm, ctx->pm, replica);
    } else {
        if ((rc = xown_price(ctx, replica, mv, 1, nullptr))) return rc;
        hipLaunchKernelGGL(k_scalar_apply, dim3(1), dim3(64), scalar_table_bytes(ctx), ctx->stream, ctx->sm, replica, mv->kind, mv->a,
                           mv->b, mv->value, ctx->d_ok);
    }
    xown_commit(ctx, replica);
    HIPCHK(ctx, hipGetLastError());
    int32_t ok = 0;
    HIPCHK(ctx, hipMemcpyAsync(&ok, ctx->d_ok, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!ok) return fail(ctx, SF_ERR_INVALID, "move is not doable");
    return SF_OK;
}

// ≙ ListCheapestInsertionPhase over every replica's current lists (csrc/sf_construct.hip)
// cheapest insertion on a list class scored by the precedence constraint (k_prec_construct_cheapest); with the slot's precedence policy
// the phase has the hooks and re-ranks the elements by their downstream chain (cheapest/kernel.rs:75-81,162-229)
static int ensure_plf(sf_ctx* ctx);
static int construct_cheapest_precedence(sf_ctx* ctx, const uint32_t* elements, int32_t n, int64_t* out_scores) {
    if (ctx->lm.dist_level >= 0 || ctx->lm.cap_level >= 0)
        return fail(ctx, SF_ERR_UNSUPPORTED, "cheapest insertion on a precedence model with distance / capacity constraints");
    if (int rc = ensure_plf(ctx)) return rc;
    std::vector<uint32_t> order(elements, elements + n);
    const PrecSpec& ps = ctx->prec;
    const size_t nodes = ps.dur.size();
    if (ctx->prec_policy && n > 0) {  // precedence_downstream: unassigned elements only (those already in a list are skipped by the kernel anyway)
        std::vector<char> in_list(nodes, 0);
        {
            std::vector<uint32_t> off((size_t)ctx->lm.V + 1), vis;
            if (hipMemcpy(off.data(), ctx->lm.off, off.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return fail(ctx, SF_ERR_HIP, "copy of the list offsets");
            vis.resize(off.back());
            if (!vis.empty() && hipMemcpy(vis.data(), ctx->lm.visits, vis.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(ctx, SF_ERR_HIP, "copy of the lists");
            for (uint32_t x : vis)
                if (x < nodes) in_list[x] = 1;
        }
        std::vector<uint32_t> el;
        for (uint32_t x : order)
            if (x < nodes && !in_list[x]) el.push_back(x);
        const size_t m = el.size();
        std::vector<int64_t> position(nodes, -1);
        bool ok = true;
        for (size_t i = 0; i < m; ++i) position[el[i]] = (int64_t)i;
        std::vector<std::vector<size_t>> succ(m);
        std::vector<size_t> preds(m, 0);
        for (size_t i = 0; i < m; ++i)
            for (uint32_t t = ps.succ_off[el[i]]; t < ps.succ_off[el[i] + 1]; ++t) {
                const uint32_t to = ps.succ[t];
                if (to >= nodes || position[to] < 0) continue;
                succ[i].push_back((size_t)position[to]);
                preds[(size_t)position[to]] += 1;
            }
        std::vector<size_t> ready, topo;
        for (size_t i = 0; i < m; ++i)
            if (preds[i] == 0) ready.push_back(i);
        while (!ready.empty()) {
            const size_t i = ready.back();
            ready.pop_back();
            topo.push_back(i);
            for (size_t s2 : succ[i])
                if (--preds[s2] == 0) ready.push_back(s2);
        }
        ok = topo.size() == m;
        if (ok) {
            std::vector<int64_t> down(m);
            for (size_t t = m; t-- > 0;) {
                const size_t i = topo[t];
                int64_t tail = 0;
                for (size_t s2 : succ[i]) tail = std::max(tail, down[s2]);
                down[i] = (int64_t)ps.dur[el[i]] + tail;
            }
            std::vector<size_t> idx(m);
            for (size_t i = 0; i < m; ++i) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return down[a] > down[b]; });
            order.clear();
            for (size_t i : idx) order.push_back(el[i]);
        }
    }
    const size_t lds = (((size_t)ctx->lm.V + 1 + 3) & ~(size_t)3) * 4 + ((((size_t)ctx->lm.dim + 31) / 32 + 3) & ~(size_t)3) * 4 + (size_t)ctx->lm.n_cap * 2 + 16;
    if (lds > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "list class does not fit one wave's LDS slice");
    uint32_t* d_el = nullptr;
    if (!order.empty()) {
        hipError_t ea = hipMalloc((void**)&d_el, order.size() * 4);
        if (ea == hipSuccess) ea = hipMemcpyAsync(d_el, order.data(), order.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (ea != hipSuccess) {
            (void)hipFree(d_el);
            return fail(ctx, SF_ERR_HIP, hipGetErrorString(ea));
        }
    }
    const int lvl_order = ctx->pm.hard_level < ctx->pm.mk_level ? 0 : (ctx->pm.hard_level > ctx->pm.mk_level ? 1 : 2);
    auto kern = k_prec_construct_cheapest;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(ctx->R), dim3(64), lds, ctx->stream, ctx->lm, ctx->pm, ctx->plf, d_el, (int)order.size(), lvl_order, ctx->sp.stats);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_el);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return run_evaluate_all(ctx, out_scores, 1);
}

int32_t sf_construct_list_cheapest(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (ctx && ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_construct_list_cheapest: a model with the join of its two planning classes is searched by the fused engine only");
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->has_list_model || descriptor_index != ctx->list_desc) return fail(ctx, SF_ERR_INVALID, "cheapest insertion needs the list variable's class");
    if (n < 0 || (n > 0 && !elements)) return fail(ctx, SF_ERR_INVALID, "bad sf_construct_list_cheapest arguments");
    if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536) return fail(ctx, SF_ERR_UNSUPPORTED, "construction packs list elements in 16 bits");
    for (int32_t k = 0; k < n; ++k)
        if (elements[k] >= (uint32_t)ctx->lm.dim) return fail(ctx, SF_ERR_INVALID, "element id out of range");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    if (ctx->pm.on) return construct_cheapest_precedence(ctx, elements, n, out_scores);
    const ConstructCarve cv(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim);
    if (cv.total > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "list class does not fit one wave's LDS slice");
    uint32_t* d_el = nullptr;
    if (n > 0) {
        hipError_t ea = hipMalloc((void**)&d_el, (size_t)n * 4);
        if (ea == hipSuccess) ea = hipMemcpyAsync(d_el, elements, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
        if (ea != hipSuccess) {
            (void)hipFree(d_el);
            return fail(ctx, SF_ERR_HIP, hipGetErrorString(ea));
        }
    }
    hipError_t e = hipSuccess;
    if (ctx->levels <= 2) {
        auto kern = k_list_construct_cheapest<2>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(ctx->R), dim3(64), cv.total, ctx->stream, ctx->lm, d_el, n, ctx->sp.stats);
    } else {
        auto kern = k_list_construct_cheapest<4>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(ctx->R), dim3(64), cv.total, ctx->stream, ctx->lm, d_el, n, ctx->sp.stats);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_el);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return run_evaluate_all(ctx, out_scores, 1);  // finish_construction: the committed score of the constructed lists
}

// ≙ ListRegretInsertionPhase over every replica's current lists (csrc/sf_construct.hip)
int32_t sf_construct_list_regret(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, const int64_t* order_keys,
                                 const int32_t* owners, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (ctx && ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_construct_list_regret: a model with the join of its two planning classes is searched by the fused engine only");
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->has_list_model || descriptor_index != ctx->list_desc) return fail(ctx, SF_ERR_INVALID, "regret insertion needs the list variable's class");
    if (n < 0 || (n > 0 && !elements)) return fail(ctx, SF_ERR_INVALID, "bad sf_construct_list_regret arguments");
    if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536 || n > 65535) return fail(ctx, SF_ERR_UNSUPPORTED, "construction packs list elements in 16 bits");
    if (ctx->pm.on) return fail(ctx, SF_ERR_UNSUPPORTED, "regret insertion on a model with precedence hooks");
    {
        std::vector<uint8_t> seen((size_t)ctx->lm.dim, 0);
        for (int32_t k = 0; k < n; ++k) {
            if (elements[k] >= (uint32_t)ctx->lm.dim) return fail(ctx, SF_ERR_INVALID, "element id out of range");
            if (seen[elements[k]]++) return fail(ctx, SF_ERR_INVALID, "duplicate element id (the source binding of the phase refuses it, regret.rs:228-236)");
            if (owners && owners[k] < -1) return fail(ctx, SF_ERR_INVALID, "owners[k]: -1 = unrestricted, otherwise the owner hook's value");
        }
    }
    int rc = alloc_search(ctx);
    if (rc) return rc;
    // the unassigned elements in (construction order key, source index) order (execute.rs:81-88); an element whose owner hook names
    // no list has no candidate entity (mod.rs:104-114) and is never placed: dropped here
    std::vector<int32_t> order;
    for (int32_t k = 0; k < n; ++k)
        if (!owners || owners[k] < ctx->lm.V) order.push_back(k);
    if (order_keys) std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return order_keys[a] < order_keys[b]; });
    const int32_t ne = (int32_t)order.size();
    std::vector<uint32_t> el((size_t)ne);
    std::vector<int32_t> ow((size_t)ne, -1);
    std::vector<uint64_t> bucket((size_t)(ctx->lm.V > 0 ? ctx->lm.V : 1), 0);
    for (int32_t k = 0; k < ne; ++k) {
        el[k] = elements[order[k]];
        if (owners) ow[k] = owners[order[k]];
        if (ow[k] >= 0) bucket[ow[k]] += 1;
    }
    if (owners) {  // kernel/fallback.rs:58-84: all-fixed-owner inputs above the trial budget take bounded fallbacks that are not built.  The
        // budget is checked on the fixed-owner elements handed over (a replica's unassigned subset can only be smaller)
        uint64_t trials = 0;
        for (uint64_t len : bucket) trials += len * (len + 1) * (len + 2) / 6;
        if (trials > 16384) return fail(ctx, SF_ERR_UNSUPPORTED, "owner-restricted regret insertion above the reference's trial budget (regret/kernel/fallback.rs)");
    }
    const RegretCarve cv(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim, ne);
    if (cv.total > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "list class does not fit one wave's LDS slice");
    uint32_t* d_el = nullptr;
    int32_t* d_ow = nullptr;
    if (ne > 0) {
        hipError_t ea = hipMalloc((void**)&d_el, (size_t)ne * 4);
        if (ea == hipSuccess) ea = hipMalloc((void**)&d_ow, (size_t)ne * 4);
        if (ea == hipSuccess) ea = hipMemcpy(d_el, el.data(), (size_t)ne * 4, hipMemcpyHostToDevice);
        if (ea == hipSuccess) ea = hipMemcpy(d_ow, ow.data(), (size_t)ne * 4, hipMemcpyHostToDevice);
        if (ea != hipSuccess) {
            (void)hipFree(d_el);
            (void)hipFree(d_ow);
            return fail(ctx, SF_ERR_HIP, hipGetErrorString(ea));
        }
    }
    hipError_t e = hipSuccess;
    if (ctx->levels <= 2) {
        auto kern = k_list_construct_regret<2>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(ctx->R), dim3(64), cv.total, ctx->stream, ctx->lm, d_el, owners ? d_ow : (const int32_t*)nullptr, ne, ctx->sp.stats);
    } else {
        auto kern = k_list_construct_regret<4>;
        e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
        if (e == hipSuccess) hipLaunchKernelGGL(kern, dim3(ctx->R), dim3(64), cv.total, ctx->stream, ctx->lm, d_el, owners ? d_ow : (const int32_t*)nullptr, ne, ctx->sp.stats);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_el);
    (void)hipFree(d_ow);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return run_evaluate_all(ctx, out_scores, 1);  // the committed score of the constructed lists
}

// ≙ ListKOptPhase (route-local 2-opt) over every replica's current lists (csrc/sf_clarke_wright.hip)
int32_t sf_construct_list_k_opt(sf_ctx* ctx, int32_t descriptor_index, int32_t k, int32_t feasible_mode, int32_t max_sweeps, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (ctx && ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_construct_list_k_opt: a model with the join of its two planning classes is searched by the fused engine only");
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->has_list_model || descriptor_index != ctx->list_desc) return fail(ctx, SF_ERR_INVALID, "list k-opt needs the list variable's class");
    if (feasible_mode != 0 && feasible_mode != 1) return fail(ctx, SF_ERR_INVALID, "feasible_mode: 0 no feasibility hook, 1 capacity");
    if (max_sweeps < 1) return fail(ctx, SF_ERR_INVALID, "max_sweeps must be >= 1 (the termination policy of the phase)");
    if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536) return fail(ctx, SF_ERR_UNSUPPORTED, "construction packs list elements in 16 bits");
    if (ctx->pm.on) return fail(ctx, SF_ERR_UNSUPPORTED, "list k-opt on a model with precedence hooks");
    if (!ctx->lm.mat) return fail(ctx, SF_ERR_UNSUPPORTED, "list k-opt needs the distance matrix (route_distance)");
    if (feasible_mode == 1 && !ctx->lm.demand) return fail(ctx, SF_ERR_INVALID, "capacity feasibility needs the demand column");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    if (k == 2 && ctx->lm.V > 0) {  // only k = 2 is implemented by the reference: every other value is a scored no-op (kernel.rs:69-77)
        const size_t lds = align_up((size_t)ctx->lm.n_cap * 2, 16) + 16;
        if (lds > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "a route does not fit one wave's LDS slice");
        hipError_t e = hipFuncSetAttribute((const void*)k_list_construct_two_opt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_list_construct_two_opt, dim3((unsigned)ctx->lm.V, (unsigned)ctx->R), dim3(64), lds, ctx->stream, ctx->lm, feasible_mode, max_sweeps, ctx->sp.stats);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    }
    return run_evaluate_all(ctx, out_scores, 1);
}

// ≙ ListConstructionPhase (round robin) over every replica's current lists (csrc/sf_construct.hip)
int32_t sf_construct_list_round_robin(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, const int64_t* order_keys,
                                      const int32_t* owners, int64_t* out_scores) {
    DeviceGuard _dev(ctx);
    if (ctx && ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_construct_list_round_robin: a model with the join of its two planning classes is searched by the fused engine only");
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->has_list_model || descriptor_index != ctx->list_desc) return fail(ctx, SF_ERR_INVALID, "round robin needs the list variable's class");
    if (n < 0 || (n > 0 && !elements)) return fail(ctx, SF_ERR_INVALID, "bad sf_construct_list_round_robin arguments");
    if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536 || n > 65535) return fail(ctx, SF_ERR_UNSUPPORTED, "construction packs list elements in 16 bits");
    std::vector<int32_t> order;
    {
        std::vector<bool> seen((size_t)ctx->lm.dim, false);
        for (int32_t k = 0; k < n; ++k) {
            if (elements[k] >= (uint32_t)ctx->lm.dim) return fail(ctx, SF_ERR_INVALID, "element id out of range");
            if (seen[elements[k]]) return fail(ctx, SF_ERR_INVALID, "duplicate element");
            seen[elements[k]] = true;
            if (owners && owners[k] < -1) return fail(ctx, SF_ERR_INVALID, "owners[k]: -1 = unrestricted, otherwise the owner hook's value");
            if (owners && owners[k] >= ctx->lm.V) continue;  // OwnerRestriction::Invalid (list_placement.rs:66-67): skipped
            order.push_back(k);
        }
    }
    if (order_keys) std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return order_keys[a] < order_keys[b]; });
    const int ne = (int)order.size();
    int rc = alloc_search(ctx);
    if (rc) return rc;
    if (ne == 0 || ctx->lm.V == 0) return run_evaluate_all(ctx, out_scores, 1);
    std::vector<uint32_t> el((size_t)ne);
    std::vector<int32_t> ow((size_t)ne, -1);
    for (int k = 0; k < ne; ++k) {
        el[k] = elements[order[k]];
        if (owners) ow[k] = owners[order[k]];
    }
    const RoundRobinCarve cv(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim, ne);
    if (cv.total > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "list class does not fit one wave's LDS slice");
    uint32_t* d_el = nullptr;
    int32_t* d_ow = nullptr;
    hipError_t e = hipMalloc((void**)&d_el, (size_t)ne * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_ow, (size_t)ne * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_el, el.data(), (size_t)ne * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ow, ow.data(), (size_t)ne * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_list_construct_round_robin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_list_construct_round_robin, dim3(ctx->R), dim3(64), cv.total, ctx->stream, ctx->lm, d_el, owners ? d_ow : (const int32_t*)nullptr, ne,
                           ctx->sp.stats);
        e = hipGetLastError();
    }
    hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
    (void)hipFree(d_el), (void)hipFree(d_ow);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return run_evaluate_all(ctx, out_scores, 1);
}

// ≙ ListClarkeWrightPhase over every replica's current lists with the stock CVRP hook bundle (csrc/sf_clarke_wright.hip)
int32_t sf_construct_list_clarke_wright(sf_ctx* ctx, int32_t descriptor_index, const uint32_t* elements, int32_t n, int32_t feasible_mode,
                                        int64_t* out_scores, int32_t* out_committed) {
    DeviceGuard _dev(ctx);
    if (ctx && ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "sf_construct_list_clarke_wright: a model with the join of its two planning classes is searched by the fused engine only");
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->has_list_model || descriptor_index != ctx->list_desc) return fail(ctx, SF_ERR_INVALID, "Clarke-Wright needs the list variable's class");
    if (n < 0 || (n > 0 && !elements)) return fail(ctx, SF_ERR_INVALID, "bad sf_construct_list_clarke_wright arguments");
    if (feasible_mode != 0 && feasible_mode != 1) return fail(ctx, SF_ERR_INVALID, "feasible_mode: 0 structural, 1 capacity");
    if (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536) return fail(ctx, SF_ERR_UNSUPPORTED, "construction packs list elements in 16 bits");
    if (ctx->pm.on) return fail(ctx, SF_ERR_UNSUPPORTED, "Clarke-Wright on a model with precedence hooks");
    if (!ctx->lm.mat) return fail(ctx, SF_ERR_UNSUPPORTED, "Clarke-Wright needs the distance matrix (savings_distance)");
    if (feasible_mode == 1 && !ctx->lm.demand) return fail(ctx, SF_ERR_INVALID, "capacity feasibility needs the demand column");
    // declared elements in source order; a duplicate source key is a binding error in the reference (runtime_list_source.rs);
    // elements whose value is the depot of the available owners are not routed (kernel.rs:83-91)
    std::vector<uint32_t> el;
    {
        std::vector<bool> seen((size_t)ctx->lm.dim, false);
        for (int32_t k = 0; k < n; ++k) {
            if (elements[k] >= (uint32_t)ctx->lm.dim) return fail(ctx, SF_ERR_INVALID, "element id out of range");
            if (seen[elements[k]]) return fail(ctx, SF_ERR_INVALID, "duplicate element");
            seen[elements[k]] = true;
            if ((int32_t)elements[k] != ctx->lm.depot) el.push_back(elements[k]);
        }
    }
    const int ne = (int)el.size();
    if (ne > 65535) return fail(ctx, SF_ERR_UNSUPPORTED, "Clarke-Wright: more than 65535 elements");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    if (out_committed) std::fill(out_committed, out_committed + ctx->R, 0);
    if (ne == 0) return run_evaluate_all(ctx, out_scores, 1);
    const CwCarve cv(ctx->lm.V, ctx->lm.n_cap, ctx->lm.dim, ne);
    if (cv.total > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "route state does not fit one wave's LDS slice");
    int monotone = 1;
    if (ctx->lm.demand) {
        std::vector<int32_t> dem((size_t)ctx->lm.dim);
        hipError_t ed = hipMemcpy(dem.data(), ctx->lm.demand, dem.size() * 4, hipMemcpyDeviceToHost);
        if (ed != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(ed));
        for (uint32_t x : el)
            if (dem[x] < 0) monotone = 0;
    }
    if (feasible_mode == 0) monotone = 1;  // no load test: every rejection is permanent
    const uint64_t P = (uint64_t)ne * (uint64_t)(ne - 1) / 2;
    uint32_t *d_el = nullptr, *d_v0 = nullptr, *d_v1 = nullptr;
    int64_t *d_k0 = nullptr, *d_k1 = nullptr;
    int32_t* d_flag = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    hipError_t e = hipMalloc((void**)&d_el, (size_t)ne * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_flag, (size_t)ctx->R * 4);
    if (e == hipSuccess && P > 0) e = hipMalloc((void**)&d_k0, P * 8);
    if (e == hipSuccess && P > 0) e = hipMalloc((void**)&d_k1, P * 8);
    if (e == hipSuccess && P > 0) e = hipMalloc((void**)&d_v0, P * 4);
    if (e == hipSuccess && P > 0) e = hipMalloc((void**)&d_v1, P * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_el, el.data(), (size_t)ne * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && P > 0) {
        hipLaunchKernelGGL(k_cw_savings, dim3((unsigned)((ne + 255) / 256), (unsigned)ne), dim3(256), 0, ctx->stream, ctx->lm, d_el, ne, d_k0, d_v0);
        e = hipGetLastError();
        if (e == hipSuccess) e = rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (size_t)P, 0, 64, ctx->stream);
        if (e == hipSuccess) e = hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16);
        if (e == hipSuccess) e = rocprim::radix_sort_pairs_desc(d_tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, (size_t)P, 0, 64, ctx->stream);
    }
    if (e == hipSuccess) {
        e = hipFuncSetAttribute((const void*)k_cw_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cv.total);
        if (e == hipSuccess)
            hipLaunchKernelGGL(k_cw_merge, dim3(ctx->R), dim3(64), cv.total, ctx->stream, ctx->lm, d_el, ne, d_v1, P, feasible_mode, monotone, d_flag,
                               (uint64_t*)nullptr);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess && out_committed) e = hipMemcpyAsync(out_committed, d_flag, (size_t)ctx->R * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
    (void)hipFree(d_el), (void)hipFree(d_flag), (void)hipFree(d_k0), (void)hipFree(d_k1), (void)hipFree(d_v0), (void)hipFree(d_v1), (void)hipFree(d_tmp);
    if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    return run_evaluate_all(ctx, out_scores, 1);  // the committed score of the constructed lists
}

// ---- search ------------------------------------------------------------------------------
int32_t sf_solver_configure(sf_ctx* ctx, const sf_solver_config* cfg) {
    if (!ctx || !cfg) return SF_ERR_INVALID;
    if (ctx->search_alloc && cfg->late_acceptance_size > ctx->sp.la_size)
        return fail(ctx, SF_ERR_INVALID, "late_acceptance_size cannot grow after the search state exists");
    if (cfg->acceptor != SF_ACCEPT_HILL_CLIMBING && cfg->acceptor != SF_ACCEPT_LATE_ACCEPTANCE &&
        cfg->acceptor != SF_ACCEPT_SIMULATED_ANNEALING && cfg->acceptor != SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE)
        return fail(ctx, SF_ERR_UNSUPPORTED, "acceptor kind");
    if (cfg->forager < SF_FORAGER_ACCEPTED_COUNT || cfg->forager > SF_FORAGER_FIRST_LAST_STEP_SCORE_IMPROVING)
        return fail(ctx, SF_ERR_UNSUPPORTED, "forager");
    if (cfg->forager == SF_FORAGER_ACCEPTED_COUNT && cfg->accepted_count_limit <= 0)
        return fail(ctx, SF_ERR_INVALID, "AcceptedCountForager: accepted_count_limit must be > 0");
    if ((cfg->acceptor == SF_ACCEPT_LATE_ACCEPTANCE || cfg->acceptor == SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE) && cfg->late_acceptance_size <= 0)
        return fail(ctx, SF_ERR_INVALID, "late_acceptance_size must be > 0");  // both acceptors assert it (late_acceptance.rs:71, diversified_late_acceptance.rs:87-90)
    ctx->cfg = *cfg;
    if (ctx->search_alloc) ctx->sp.la_size = cfg->late_acceptance_size > 0 ? cfg->late_acceptance_size : 1;
    return SF_OK;
}

// compile_default_local_search_components (runtime/compiler/default_local_search/policy.rs:21-82)
int32_t sf_default_local_search_components(int32_t has_lists, int32_t has_groups, int32_t has_precedence, int32_t has_nearby_scalar,
                                           int32_t has_conflict_repairs, uint64_t random_seed, sf_solver_config* out) {
    if (!out) return SF_ERR_INVALID;
    sf_solver_config c{};
    c.late_acceptance_size = 400;  // DEFAULT_LOCAL_SEARCH_LATE_ACCEPTANCE_SIZE (:18)
    c.acceptor = has_lists ? SF_ACCEPT_LATE_ACCEPTANCE : has_groups ? SF_ACCEPT_DIVERSIFIED_LATE_ACCEPTANCE : SF_ACCEPT_SIMULATED_ANNEALING;
    if (has_groups && !has_lists) {
        c.forager = SF_FORAGER_FIRST_LAST_STEP_SCORE_IMPROVING;
        c.accepted_count_limit = 0;  // accepted_count_limit: None
    } else if (has_precedence) {
        c.forager = SF_FORAGER_FIRST_LAST_STEP_SCORE_IMPROVING;
        c.accepted_count_limit = 256;  // DEFAULT_LOCAL_SEARCH_ACCEPTED_COUNT (:19)
    } else {
        c.forager = SF_FORAGER_ACCEPTED_COUNT;
        c.accepted_count_limit = (has_lists || has_nearby_scalar || has_conflict_repairs) ? 256 : 1;
    }
    c.random_ties = 1;
    c.selection_order = SF_ORDER_RANDOM;  // every default leaf is compiled with SelectionOrder::Random (:118)
    c.random_seed = random_seed;
    *out = c;
    return SF_OK;
}

int32_t sf_provider_declare(sf_ctx* ctx, int32_t kind, const char* name) {
    if (!ctx) return SF_ERR_INVALID;
    if (kind != SF_PROVIDER_SCALAR_GROUP && kind != SF_PROVIDER_CONFLICT_REPAIR) return fail(ctx, SF_ERR_INVALID, "sf_provider_declare: unknown provider kind");
    if (!name || !*name) return fail(ctx, SF_ERR_INVALID, "sf_provider_declare: a provider needs a name (the group's / the repaired constraint's)");
    ctx->providers.push_back({kind, std::string(name)});
    return SF_OK;
}

int32_t sf_solver_configure_default(sf_ctx* ctx, uint64_t random_seed, int32_t has_groups, int32_t has_conflict_repairs, sf_solver_config* out) {
    if (!ctx) return SF_ERR_INVALID;
    if (has_groups < 0 || has_conflict_repairs < 0) {  // derived from what was declared
        bool g = false, r = false;
        for (const auto& pv : ctx->providers) {
            g = g || pv.first == SF_PROVIDER_SCALAR_GROUP;
            r = r || pv.first == SF_PROVIDER_CONFLICT_REPAIR;
        }
        if (has_groups < 0) has_groups = g ? 1 : 0;
        if (has_conflict_repairs < 0) has_conflict_repairs = r ? 1 : 0;
    }
    bool has_lists = false, has_precedence = ctx->prec_policy, has_nearby_scalar = false;
    for (const auto& kv : ctx->classes) has_lists = has_lists || kv.second.has_list;
    for (const auto& s : ctx->selectors) {
        has_precedence = has_precedence || s.kind == SF_SEL_LIST_PRECEDENCE;
        has_nearby_scalar = has_nearby_scalar || s.kind == SF_SEL_NEARBY_SCALAR_CHANGE || s.kind == SF_SEL_NEARBY_SCALAR_SWAP;
    }
    sf_solver_config c{};
    sf_default_local_search_components(has_lists, has_groups, has_lists && has_precedence, has_nearby_scalar, has_conflict_repairs, random_seed, &c);
    const int32_t rc = sf_solver_configure(ctx, &c);
    if (rc == SF_OK && out) *out = c;
    return rc;
}

// DiversifiedLateAcceptanceAcceptor::new(late_acceptance_size, tolerance) (diversified_late_acceptance.rs:86-98); the history
// size is sf_solver_config::late_acceptance_size
int32_t sf_solver_configure_diversified(sf_ctx* ctx, double tolerance) {
    if (!ctx) return SF_ERR_INVALID;
    if (!std::isfinite(tolerance)) return fail(ctx, SF_ERR_INVALID, "diversified late acceptance: tolerance must be finite");
    ctx->dla_tolerance = tolerance;
    return SF_OK;
}

// assert_simulated_annealing_parameters (simulated_annealing.rs:305-336): the reference panics, the C ABI reports
int32_t sf_solver_configure_annealing(sf_ctx* ctx, const sf_annealing_config* cfg) {
    if (!ctx || !cfg) return SF_ERR_INVALID;
    auto temperature_ok = [](double v) { return std::isfinite(v) && v >= 0.0; };
    if (cfg->mode < SF_ANNEAL_SINGLE || cfg->mode > SF_ANNEAL_CALIBRATED) return fail(ctx, SF_ERR_INVALID, "annealing mode");
    if (!(std::isfinite(cfg->decay_rate) && cfg->decay_rate > 0.0 && cfg->decay_rate <= 1.0))
        return fail(ctx, SF_ERR_INVALID, "simulated_annealing decay_rate must be finite and in (0, 1]");
    if (!temperature_ok(cfg->hill_climbing_temperature))
        return fail(ctx, SF_ERR_INVALID, "simulated_annealing hill_climbing_temperature must be finite and non-negative");
    const int nt = cfg->mode == SF_ANNEAL_SINGLE ? 1 : cfg->mode == SF_ANNEAL_PER_LEVEL ? ctx->levels : 0;
    for (int k = 0; k < nt; ++k)
        if (!temperature_ok(cfg->temperatures[k]))
            return fail(ctx, SF_ERR_INVALID, "simulated_annealing level_temperatures must be finite and non-negative");
    if (cfg->mode == SF_ANNEAL_CALIBRATED) {
        if (cfg->calibration_sample_size <= 0)
            return fail(ctx, SF_ERR_INVALID, "simulated_annealing calibration sample_size must be greater than 0");
        if (!(std::isfinite(cfg->target_acceptance_probability) && cfg->target_acceptance_probability > 0.0 &&
              cfg->target_acceptance_probability < 1.0))
            return fail(ctx, SF_ERR_INVALID, "simulated_annealing calibration target_acceptance_probability must be in (0, 1)");
        if (!temperature_ok(cfg->fallback_temperature))
            return fail(ctx, SF_ERR_INVALID, "simulated_annealing calibration fallback_temperature must be finite and non-negative");
    }
    ctx->anneal = *cfg;
    ctx->anneal_seed_set = true;
    return SF_OK;
}

// acceptor.phase_started (simulated_annealing.rs:377-415) for every replica; rng = SmallRng::seed_from_u64(seed + r)
static int anneal_phase_start(sf_ctx* ctx) {
    const sf_annealing_config& a = ctx->anneal;
    SearchParams& p = ctx->sp;
    p.sa.decay_rate = a.decay_rate;
    p.sa.hill_climbing_temperature = a.hill_climbing_temperature;
    p.sa.denominator = -std::log(a.target_acceptance_probability);
    p.sa.fallback_temperature = a.fallback_temperature;
    p.sa.sample_size = a.calibration_sample_size > 0 ? a.calibration_sample_size : 1;
    p.sa.never_accept_hard = a.never_accept_hard_regression;
    p.sa.hard_levels = ctx->hard_levels;
    p.sa.levels = ctx->levels;
    const uint64_t seed = ctx->anneal_seed_set ? a.seed : ctx->cfg.random_seed;
    std::vector<uint64_t> st((size_t)ctx->R * SA_WORDS, 0);
    for (int r = 0; r < ctx->R; ++r) {
        uint64_t* w = st.data() + (size_t)r * SA_WORDS;
        uint64_t state = seed + (uint64_t)r;
        for (int i = 0; i < 4; ++i) {  // xoshiro256++ seed_from_u64: splitmix64 expansion
            state += 0x9E3779B97F4A7C15ULL;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            w[SA_RNG + i] = z ^ (z >> 31);
        }
        for (int k = 0; k < ctx->levels; ++k) {
            double t = a.mode == SF_ANNEAL_SINGLE ? a.temperatures[0] : a.mode == SF_ANNEAL_PER_LEVEL ? a.temperatures[k] : 0.0;
            std::memcpy(&w[SA_TEMP + k], &t, 8);
        }
        w[SA_CALIBRATING] = a.mode == SF_ANNEAL_CALIBRATED ? 1 : 0;
    }
    HIPCHK(ctx, hipMemcpyAsync(p.sa.state, st.data(), st.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

int32_t sf_get_annealing_state(sf_ctx* ctx, int32_t replica, double* out_temperatures, int32_t* out_calibrating) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->search_alloc || replica < 0 || replica >= ctx->R || !out_temperatures || !out_calibrating)
        return fail(ctx, SF_ERR_INVALID, "bad sf_get_annealing_state");
    uint64_t w[SA_WORDS];
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(w, ctx->sp.sa.state + (size_t)replica * SA_WORDS, sizeof(w), hipMemcpyDeviceToHost));
    for (int k = 0; k < ctx->levels; ++k) std::memcpy(&out_temperatures[k], &w[SA_TEMP + k], 8);
    *out_calibrating = (int32_t)w[SA_CALIBRATING];
    return SF_OK;
}

int32_t sf_solver_set_engine(sf_ctx* ctx, int32_t engine) {
    if (!ctx || engine < SF_ENGINE_AUTO || engine > SF_ENGINE_WAVE) return fail(ctx, SF_ERR_INVALID, "bad engine");
    if (engine == SF_ENGINE_WAVE && ctx->initialized && !wave_engine_possible(ctx))
        return fail(ctx, SF_ERR_UNSUPPORTED, "wave engine cannot run this model");
    ctx->engine = engine;
    return SF_OK;
}

int32_t sf_solver_get_engine(sf_ctx* ctx, int32_t* out_engine) {
    if (!ctx || !ctx->initialized || !out_engine) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    *out_engine = (ctx->has_scalar_model || use_wave_engine(ctx)) ? SF_ENGINE_WAVE : SF_ENGINE_BLOCK;
    return SF_OK;
}

int32_t sf_list_wave_layout(sf_ctx* ctx, int32_t* out_mode, int32_t* out_renumbered) {
    if (!ctx || !out_mode || !out_renumbered) return fail(ctx, SF_ERR_INVALID, "sf_list_wave_layout: null argument");
    *out_mode = ctx->last_wave_mode;
    *out_renumbered = ctx->last_wave_mode >= 3 && ctx->wave_renumbered ? 1 : 0;
    return SF_OK;
}

int32_t sf_solver_set_step_seeds(sf_ctx* ctx, const uint64_t* seeds, int64_t n_steps) {
    DeviceGuard _dev(ctx);
    if (!ctx) return SF_ERR_INVALID;
    if (!seeds || n_steps <= 0) {
        ctx->d_explicit = nullptr;
        ctx->n_explicit = 0;
        return SF_OK;
    }
    uint64_t* d = nullptr;
    int rc = upload(ctx, &d, seeds, (size_t)n_steps * ctx->R);
    if (rc) return rc;
    if (ctx->d_explicit) {  // release the previous sequence (no launch is using it: calls on a ctx are not re-entrant)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        for (auto it = ctx->allocs.begin(); it != ctx->allocs.end(); ++it)
            if (*it == (void*)ctx->d_explicit) {
                ctx->allocs.erase(it);
                break;
            }
        (void)hipFree(ctx->d_explicit);
    }
    ctx->d_explicit = d;
    ctx->n_explicit = n_steps;
    return SF_OK;
}

int32_t sf_phase_start(sf_ctx* ctx) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->sp.seed_draws, 0, (size_t)ctx->R * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->sp.stats, 0, (size_t)ctx->R * SF_STATS_WORDS * 8, ctx->stream));
    if ((rc = anneal_phase_start(ctx))) return rc;
    if ((rc = ruin_phase_start(ctx))) return rc;
    if (ctx->has_list_model)
        hipLaunchKernelGGL(k_list_phase_start, dim3(ctx->R), dim3(256), 0, ctx->stream, ctx->lm, ctx->sp);
    if (ctx->has_scalar_model)  // mixed: same committed score (aliased), adds the best snapshot of the values
        hipLaunchKernelGGL(k_scalar_phase_start, dim3(ctx->R), dim3(256), 0, ctx->stream, ctx->sm, ctx->sp);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

}  // extern "C"

// generic N-leaf engine: mixed models, and list models whose union has plain list change / swap leaves
template <int L, class VT, bool RUIN = false, bool PREC = false>
static int launch_mixed_t(sf_ctx* ctx, const SearchParams& p, const GLeaves& gl, int n_replicas, bool trace) {
    const int ns = ctx->has_scalar_model ? ctx->sm.n : 0;
    const bool tables = ctx->has_scalar_model && ctx->sm.tables();
    // FAST instantiation: the reference's default list policy on a list-only model (see k_mixed_search_wave)
    static const bool no_fast = std::getenv("SF_AMD_MIXED_NO_FAST") != nullptr;  // diagnostics / parity tests: force the general instantiation
    bool fast_kinds = true;  // the leaf kinds the FAST instantiation keeps (the default list policy of a slot with a distance meter)
    for (int l = 0; l < gl.n; ++l) {
        const int k = gl.kind[l];
        fast_kinds = fast_kinds && (k == 16 || k == 32 || k == 64 || k == 128 || k == 256 || k == 1024 || (k == 512 && gl.kopt_nearby));
    }
    const bool fast = !no_fast && !trace && !PREC && sizeof(VT) == 2 && ctx->has_list_model && !ctx->has_scalar_model && p.acceptor == SF_ACCEPT_LATE_ACCEPTANCE &&
                      p.forager == SF_FORAGER_ACCEPTED_COUNT && !p.dry_run && !gl.union_custom && gl.union_order == SF_UNION_STRATIFIED_RANDOM && gl.n > 1 &&
                      (ctx->lm.mat_symmetric || ctx->lm.dist_level < 0) && !p.legacy_eval && !p.explicit_seeds && fast_kinds &&
                      p.order == SF_ORDER_RANDOM &&  // (the default policy's SelectionOrder: compiled in, see StreamCtx in the kernel)
                      // with a ruin leaf the FAST kernel carries the list-preserving recreate only (sf_ruin_v2.h: rv2_model_ok + the edge table)
                      (!RUIN || (ctx->lm.leg16 && ctx->lm.V <= 128 && ctx->lm.n_cap <= 32767 && ctx->lm.dim <= 32767 && ctx->lm.small32 && ctx->lm.mat16));
    // (FAST + ruin: the list-preserving recreate only and the node -> slot table in HBM, see the kernel)
    const bool nodeg = fast && (RUIN || SF_MIXED_FAST_NODEG != 0);
    GCarve<VT> cv(ns, ctx->has_list_model ? ctx->lm.V : 0, ctx->has_list_model ? ctx->lm.n_cap : 0, gl.has_nearby ? ctx->lm.dim : 0,
                  gl.kopt_nearby, gl.n, gl.has_ruin ? (nodeg ? 3 : (ctx->lm.leg16 ? 2 : 1)) : 0, ctx->has_list_model ? ctx->lm.dim : 0,
                  PREC && gl.prec_lds ? gl.prec.n : 0, tables ? ctx->sm.n_values : 0, tables && ctx->sm.run_level >= 0 ? ctx->sm.run_P : 0,
                  PREC && gl.prec_lds ? gl.prec_groups : 0, nodeg);
    if (cv.total > SF_LDS_BUDGET) return fail(ctx, SF_ERR_UNSUPPORTED, "model does not fit one wave's LDS slice");
    GLeaves gl2 = gl;
    gl2.ringd = nullptr;
    static const bool no_pre_eval = std::getenv("SF_AMD_MIXED_NO_PRE_EVAL") != nullptr;  // diagnostics / parity tests: score inside the replay as before
    if (fast && ctx->lm.small32 && !no_pre_eval && !SF_MIXED_RING_LDS) {  // the scoring stage stores 32-bit deltas
        if (!ctx->d_mixed_ringd) {
            int32_t* rd = nullptr;
            int rc = dalloc(ctx, &rd, (size_t)ctx->R * GL * GRC * 2);
            if (rc) return rc;
            ctx->d_mixed_ringd = rd;
        }
        gl2.ringd = ctx->d_mixed_ringd;
    }
    if (nodeg) {
        if (!ctx->d_node_tab32) {
            uint32_t* nt = nullptr;
            int rc = dalloc(ctx, &nt, (size_t)ctx->R * ctx->lm.dim);
            if (rc) return rc;
            ctx->d_node_tab32 = nt;
        }
        gl2.node_tab = ctx->d_node_tab32;
    }
    // replicas (waves) per workgroup: the count that keeps the most waves resident per CU (a workgroup's LDS is
    // allocated as a whole; the kernel is built for SF_MIXED_BLOCKS_PER_CU workgroups of 4 waves per CU, the FAST
    // instantiation for SF_MIXED_FAST_BLOCKS_PER_CU); ties go to the larger group
    // precedence models: the four-workgroups-per-CU build when the launch has more replicas than two workgroups per CU hold and the LDS
    // slice lets more than eight share a CU (sf_mixed_wave.hip: MODE 2)
    static const bool no_prec_occ = std::getenv("SF_AMD_PREC_NO_OCC") != nullptr;
    // (not with the grouped evaluator: its scratch leaves room for 8 - 10 replicas per CU, and the 128-register build is slower per wave:
    // nine-leaf policy 20 x 10 at 6,144 replicas 71 M moves/s with it, 106 M without)
    const bool prec_occ = PREC && !trace && !no_prec_occ && !gl.prec_groups && n_replicas > 8 * 256 && (160 * 1024) / (cv.total + 256) > 8;
    const size_t max_waves = 4 * (size_t)(fast ? (RUIN ? SF_MIXED_FAST_RUIN_BLOCKS_PER_CU : SF_MIXED_FAST_BLOCKS_PER_CU) : (prec_occ ? SF_MIXED_PREC_BLOCKS_PER_CU : SF_MIXED_BLOCKS_PER_CU));  // by register budget
    int wpb = 1;
    size_t best_resident = 0;
    const char* wenv = std::getenv("SF_AMD_MIXED_WPB");  // diagnostics: cap the replicas per workgroup (A/B of the workgroup shape)
    const int wmax = wenv && std::atoi(wenv) >= 1 && std::atoi(wenv) < 4 ? std::atoi(wenv) : 4;
    for (int w = 1; w <= wmax; ++w) {
        const size_t per_wg = cv.total * w + (fast ? 0 : 1024) + (PREC ? (size_t)gl.prec_static : 0);  // + the static annealing state (the FAST kernels have none), the shared copy of the precedence graph
        if (per_wg > 160 * 1024) break;
        size_t groups = (160 * 1024) / per_wg;
        if (groups * w > max_waves) groups = max_waves / w;
        if (groups * w >= best_resident) {
            best_resident = groups * w;
            wpb = w;
        }
    }
    if (best_resident == 0) return fail(ctx, SF_ERR_UNSUPPORTED, "generic engine: one replica's LDS slice (with the precedence scratch / static copy) exceeds a CU's 160 KiB");
    static const bool dbg_launch = std::getenv("SF_AMD_DEBUG_LAUNCH") != nullptr;  // diagnostics: the launch shape, once per change
    if (dbg_launch) {
        static size_t last = 0;
        const size_t key = cv.total * 131 + (size_t)wpb * 7 + (size_t)n_replicas + (prec_occ ? 1 : 0);
        if (key != last) {
            last = key;
            std::fprintf(stderr, "[sf] generic engine launch: L=%d ruin=%d prec=%d fast=%d prec_occ=%d replicas=%d LDS/replica=%zu B (prec groups %d, static %d B) waves/workgroup=%d resident/CU=%zu\n",
                         L, (int)RUIN, (int)PREC, (int)fast, (int)prec_occ, n_replicas, cv.total, PREC ? gl.prec_groups : 0, PREC ? gl.prec_static : 0, wpb, best_resident);
        }
    }
    SearchParams q = p;
    q.n_launch = n_replicas;
    HIPCHK(ctx, (launch_tu_mixed<L, (int)sizeof(VT), RUIN, PREC>(trace, fast ? 1 : (prec_occ ? 2 : 0), make_launch(ctx, &q, (n_replicas + wpb - 1) / wpb, 64 * wpb, cv.total * wpb + (PREC ? (size_t)gl.prec_static : 0), &gl2))));
    return SF_OK;
}
static bool has_plain_list_leaves(sf_ctx* ctx) {
    for (auto& s : ctx->selectors)
        if (s.desc == ctx->list_desc && (s.kind == SF_SEL_LIST_CHANGE || s.kind == SF_SEL_LIST_SWAP || s.kind == SF_SEL_LIST_REVERSE ||
                                         s.kind == SF_SEL_SUBLIST_CHANGE || s.kind == SF_SEL_SUBLIST_SWAP || s.kind == SF_SEL_KOPT ||
                                         s.kind == SF_SEL_LIST_RUIN || s.kind == SF_SEL_LIST_PERMUTE || s.kind == SF_SEL_LIST_PRECEDENCE))
            return true;
    return false;
}
// per-replica tables of the critical-path leaf / the route-graph filter / the precedence-aware recreate (sf_prec_leaf.h)
static int ensure_plf(sf_ctx* ctx) {
    if (ctx->plf.on) return SF_OK;
    const PrecSpec& ps = ctx->prec;
    const size_t n = ps.dur.size();
    std::vector<int32_t> deg(n, 0);
    for (size_t i = 0; i < n; ++i) {
        std::vector<uint32_t> seen;
        for (uint32_t t = ps.succ_off[i]; t < ps.succ_off[i + 1]; ++t) {
            const uint32_t to = ps.succ[t];
            if (to >= n) continue;
            if (std::find(seen.begin(), seen.end(), to) != seen.end())
                return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence leaf: a node names one fixed successor twice");
            seen.push_back(to);
            deg[i] += 1, deg[to] += 1;
        }
    }
    PlfModel& pl = ctx->plf;
    pl.dmax = 0;
    for (int32_t dv : deg) pl.dmax = std::max(pl.dmax, dv);
    // flag / first serve as per-position tables (analysis) and as per-node tables (recreate, construction): rows of max(n, n_cap) words
    const size_t R = (size_t)ctx->R, nn = std::max<size_t>(n, 1), nc = std::max<size_t>(nn, (size_t)std::max(ctx->lm.n_cap, 1));
    pl.pc = (int32_t)nc;
    int rc = dalloc(ctx, &pl.latest, R * nn);
    if (!rc) rc = dalloc(ctx, &pl.posn, R * nn);
    if (!rc) rc = dalloc(ctx, &pl.flag, R * nc);
    if (!rc) rc = dalloc(ctx, &pl.roff, R * (nn + 2));
    if (!rc) rc = dalloc(ctx, &pl.blk, R * nn * 2);
    if (!rc) rc = dalloc(ctx, &pl.csw, R * nn);
    if (!rc) rc = dalloc(ctx, &pl.ssw, R * nn);
    if (!rc) rc = dalloc(ctx, &pl.first, R * nc);
    if (!rc) rc = dalloc(ctx, &pl.cnl, R * nn);
    if (!rc) rc = dalloc(ctx, &pl.msrow, R * (nn + 1));
    if (!rc) rc = dalloc(ctx, &pl.mrrow, R * (nn + 1));
    if (!rc) rc = dalloc(ctx, &pl.sE, R * (size_t)std::max(ctx->lm.V, 1));
    if (!rc) rc = dalloc(ctx, &pl.score, R * GRC * 4);
    if (!rc) rc = dalloc(ctx, &pl.visit, R * nn);
    if (!rc) rc = dalloc(ctx, &pl.cache, R * GL * GRC * 2);
    if (rc) return rc;
    pl.on = 1;
    return SF_OK;
}
static int launch_mixed(sf_ctx* ctx, SearchParams& p, int grid, bool trace) {
    GLeaves gl{};
    gl.list_desc = ctx->has_list_model ? ctx->list_desc : 0;
    gl.xown_level = ctx->xown_level, gl.xown_weight = ctx->xown_weight, gl.xown_tab = ctx->d_xown_tab;
    // default-policy declaration order: list rules first, then scalar change, scalar swap
    // (runtime/compiler/default_local_search/policy.rs:104-108).  A configured root union (sf_union_configure) keeps the
    // order of the sf_selector_add calls instead: its weights and the Sequential / RoundRobin child order follow the
    // declaration order of the UnionMoveSelectorConfig's children (vec_union.rs:119-124).
    std::vector<const SelectorSpec*> ordered;
    if (union_is_custom(ctx)) {
        for (auto& s : ctx->selectors) ordered.push_back(&s);
    } else {
        for (int kind : {SF_SEL_LIST_PRECEDENCE, SF_SEL_LIST_PERMUTE,  // the precedence pair leads the list policy (policy/list.rs:24-33)
                         SF_SEL_NEARBY_LIST_CHANGE, SF_SEL_LIST_CHANGE, SF_SEL_NEARBY_LIST_SWAP, SF_SEL_LIST_SWAP, SF_SEL_SUBLIST_CHANGE,
                         SF_SEL_SUBLIST_SWAP, SF_SEL_LIST_REVERSE, SF_SEL_KOPT, SF_SEL_LIST_RUIN, SF_SEL_NEARBY_SCALAR_CHANGE,
                         SF_SEL_NEARBY_SCALAR_SWAP, SF_SEL_SCALAR_CHANGE, SF_SEL_SCALAR_SWAP})  // nearby scalar rules precede the ordinary pair (policy.rs:104-108)
            for (auto& s : ctx->selectors)
                if (s.kind == kind) ordered.push_back(&s);
    }
        for (const SelectorSpec* sp : ordered) {
            const SelectorSpec& s = *sp;
            const int kind = s.kind;
            const bool is_list = kind != SF_SEL_SCALAR_CHANGE && kind != SF_SEL_SCALAR_SWAP && kind != SF_SEL_NEARBY_SCALAR_CHANGE && kind != SF_SEL_NEARBY_SCALAR_SWAP;
            if (is_list ? (!ctx->has_list_model || s.desc != ctx->list_desc) : (!ctx->has_scalar_model || s.desc != ctx->scalar_desc))
                continue;
            if (gl.n >= GL) return fail(ctx, SF_ERR_UNSUPPORTED, "too many leaves for the generic engine");
            if (kind == SF_SEL_NEARBY_LIST_CHANGE || kind == SF_SEL_NEARBY_LIST_SWAP) {
                if (!wave_engine_possible(ctx)) return fail(ctx, SF_ERR_UNSUPPORTED, "nearby leaves need the neighbour index (matrix meter, <= 16384 nodes)");
                if (gl.has_nearby >= 2) return fail(ctx, SF_ERR_UNSUPPORTED, "at most two nearby leaves per union");
                gl.has_nearby += 1;
                gl.max_nearby[gl.n] = s.max_nearby;
            }
            if (kind == SF_SEL_NEARBY_SCALAR_CHANGE || kind == SF_SEL_NEARBY_SCALAR_SWAP) {
                const int which = kind == SF_SEL_NEARBY_SCALAR_CHANGE ? 0 : 1;
                NearbyScalarSource& src = ctx->nearby_scalar[which];
                if (!src.d_off) {
                    int rc = upload(ctx, &src.d_off, src.off.data(), src.off.size());
                    if (!rc) rc = upload(ctx, &src.d_val, src.val.data(), src.val.size());
                    if (rc) return rc;
                }
                gl.ns_off[which] = src.d_off;
                gl.ns_val[which] = src.d_val;
                gl.ns_dynamic = ctx->nearby_scalar_dynamic;
                gl.max_nearby[gl.n] = s.max_nearby;
            }
            if (kind == SF_SEL_KOPT) {
                for (int l = 0; l < gl.n; ++l)
                    if (gl.kind[l] == SF_SEL_KOPT) return fail(ctx, SF_ERR_UNSUPPORTED, "one 3-opt leaf per union");
                gl.max_nearby[gl.n] = s.max_nearby;
                if (s.max_nearby > 0) {
                    if (!ctx->lm.mat) return fail(ctx, SF_ERR_INVALID, "distance-pruned 3-opt needs the matrix meter");
                    if (!ctx->d_kopt_scratch) {
                        int rc = dalloc(ctx, &ctx->d_kopt_scratch, (size_t)ctx->R * ctx->lm.n_cap);
                        if (rc) return rc;
                    }
                    gl.kopt_nearby = 1;
                    gl.kopt_scratch = ctx->d_kopt_scratch;
                }
            }
            if (kind == SF_SEL_LIST_PRECEDENCE) {  // critical-path leaf: per-replica tables (sf_prec_leaf.h)
                if (!ctx->pm.on) return fail(ctx, SF_ERR_INVALID, "list precedence leaf: the list class carries no precedence constraint");
                // its ruins recreate by the precedence constraint alone: no other list constraint may score an insertion
                if (ctx->lm.dist_level >= 0 || ctx->lm.cap_level >= 0)
                    return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence leaf on a list class with distance / capacity constraints");
                // the multi-swap stream (critical x critical x support triples, < nodes^3 / 2) is indexed in 64 bits (round 4; 62 of them travel in a
                // ring entry); list positions and block lengths are 16-bit fields
                if (ctx->prec.dur.size() > 65535) return fail(ctx, SF_ERR_UNSUPPORTED, "list precedence leaf: more than 65,535 nodes");
                if (int rc = ensure_plf(ctx)) return rc;
                gl.plf = ctx->plf;
                gl.plf.leaf = 1;
            }
            if (kind == SF_SEL_LIST_PRECEDENCE && ctx->xown_level >= 0)
                return fail(ctx, SF_ERR_UNSUPPORTED, "the join of the two planning classes is not priced by the critical-path leaf's block moves");
            if (kind == SF_SEL_LIST_RUIN) {
                if (gl.has_ruin) return fail(ctx, SF_ERR_UNSUPPORTED, "one list ruin leaf per union");
                if (ctx->xown_level >= 0) return fail(ctx, SF_ERR_UNSUPPORTED, "the join of the two planning classes is not priced by the ruin leaf's recreate");
                if (!ctx->d_ruin_rng) return fail(ctx, SF_ERR_INVALID, "list ruin leaf: sf_phase_start seeds its stream first");
                gl.has_ruin = 1;
                gl.ruin = RuinParams{s.min_size, s.max_size, s.moves_per_step, s.max_source_len, s.skip_empty, ctx->d_ruin_rng};
            }
            gl.min_size[gl.n] = s.min_size;
            gl.max_size[gl.n] = s.max_size;
            gl.kind[gl.n++] = kind;
        }
    if (gl.n == 0) return fail(ctx, SF_ERR_INVALID, "no selector configured");
    if (ctx->has_list_model && ctx->pm.on && (ctx->prec_policy || gl.has_ruin)) {  // route-graph filter / recreate scored by the precedence constraint
        if (int rc = ensure_plf(ctx)) return rc;
        const int leaf = gl.plf.leaf;
        gl.plf = ctx->plf;
        gl.plf.leaf = leaf;
        gl.plf.policy = ctx->prec_policy ? 1 : 0;
    }
    gl.plf.slow = std::getenv("SF_AMD_PLF_SLOW") != nullptr ? 1 : 0;  // read at every launch
    gl.plf.force64 = std::getenv("SF_AMD_PLF_FORCE64") != nullptr ? 1 : 0;
    if (!ctx->d_mixed_ring) {
        int rc = dalloc(ctx, &ctx->d_mixed_ring, (size_t)ctx->R * GL * GRC * 2);
        if (!rc) rc = dalloc(ctx, &ctx->d_mixed_ringx, (size_t)ctx->R * GL * GRC);
        if (rc) return rc;
    }
    gl.ring = ctx->d_mixed_ring;
    gl.ringx = ctx->d_mixed_ringx;
    gl.union_order = ctx->union_order >= 0 ? ctx->union_order : (gl.n > 1 ? SF_UNION_STRATIFIED_RANDOM : SF_UNION_SEQUENTIAL);
    gl.union_custom = union_is_custom(ctx) && gl.n > 1 ? 1 : 0;
    for (int l = 0; l < GL; ++l) gl.weight[l] = 1;
    if (!ctx->union_weights.empty()) {
        if ((int)ctx->union_weights.size() != gl.n) return fail(ctx, SF_ERR_INVALID, "union weight count must match child count");
        for (int l = 0; l < gl.n; ++l) gl.weight[l] = ctx->union_weights[l];
    }
    if (ctx->has_list_model && (ctx->lm.n_cap > 65535 || ctx->lm.dim > 65536))
        return fail(ctx, SF_ERR_UNSUPPORTED, "generic engine packs list elements and positions in 16 bits");
    p.n_leaves = gl.n;
    // two level counts (2, 4); i16 values, and i8 values for models whose scalar class dominates the LDS slice (a
    // replica's value array in one byte per entity: job shop 500 x 20 fits 4 waves per CU instead of 3)
    gl.levels = ctx->levels;
    gl.prec = ctx->has_list_model ? ctx->pm : PrecModel{};
    {  // the Kahn scratch (16 bytes per node) goes to LDS while at least 4 replicas still fit a CU
        const bool no_lds = std::getenv("SF_AMD_PREC_HBM") != nullptr;  // diagnostics / parity tests: force the HBM scratch (read at every launch)
        // ... and beyond that whenever ONE replica per CU still fits: 10,000 nodes (job shop 500 x 20) run 1.7 x the rate of the HBM scratch with
        // half the replicas (profiles/r05_prec_eval_ab.txt).  SF_AMD_PREC_LDS_MAX_KB caps the scratch (36 = the round-4 rule)
        size_t lds_max = SF_LDS_BUDGET;
        if (const char* e = std::getenv("SF_AMD_PREC_LDS_MAX_KB")) lds_max = (size_t)std::atoi(e) * 1024;
        bool fits = gl.prec.on && !no_lds && prec_lds_scratch_bytes(gl.prec.n) <= lds_max && gl.prec.n < 65535;
        if (fits && prec_lds_scratch_bytes(gl.prec.n) > 36 * 1024) {  // the whole slice of a replica with the scratch in it (2-byte values: the larger carve)
            const int ns = ctx->has_scalar_model ? ctx->sm.n : 0;
            const bool tables = ctx->has_scalar_model && ctx->sm.tables();
            const GCarve<int16_t> cv(ns, ctx->lm.V, ctx->lm.n_cap, gl.has_nearby ? ctx->lm.dim : 0, gl.kopt_nearby, gl.n, gl.has_ruin ? (ctx->lm.leg16 ? 2 : 1) : 0,
                                     ctx->lm.dim, gl.prec.n, tables ? ctx->sm.n_values : 0, tables && ctx->sm.run_level >= 0 ? ctx->sm.run_P : 0, 0, false);
            fits = cv.total + 1024 <= SF_LDS_BUDGET;
        }
        gl.prec_lds = fits ? 1 : 0;
        // the incremental trial refresh is parity-complete but SLOWER than one full evaluation per trial on every job shop measured
        // (profiles/r03f_precedence.txt): opt-in for the parity tests and further work
        gl.prec_inc = std::getenv("SF_AMD_PREC_INC") != nullptr ? 1 : 0;
        // lane-per-trial sweep (prec_trial_sweep64): the default with the scratch in HBM; SF_AMD_PREC_NO_SWEEP = one full evaluation per trial
        // the constraint's static graph (durations, fixed successors / predecessors, in-degrees, owners) once per workgroup in LDS: every Kahn
        // round reads it behind a dependent LDS access (SF_AMD_PREC_STATIC_HBM = leave it in HBM / L1)
        gl.prec_static = 0, gl.prec_static_slim = 0;
        if (gl.prec.on && gl.prec_lds && std::getenv("SF_AMD_PREC_STATIC_HBM") == nullptr) {
            const size_t b = prec_static_bytes(gl.prec.n, gl.prec.n_edges, gl.prec.owner != nullptr);
            if (b <= 16 * 1024) gl.prec_static = (int32_t)b;
            // beyond that: the node records, fixed in-degrees and owners alone (what every evaluation reads per node; the rounds of a 1,000-node
            // evaluation waited on an L2 round trip for the record otherwise).  SF_AMD_PREC_STATIC_SLIM=0 leaves them in HBM
            const size_t sb = prec_static_slim_bytes(gl.prec.n, gl.prec.owner != nullptr);
            const char* se = std::getenv("SF_AMD_PREC_STATIC_SLIM");
            if (!gl.prec_static && sb <= 40 * 1024 && !(se && std::atoi(se) == 0)) gl.prec_static = (int32_t)sb, gl.prec_static_slim = 1;
            // the shared copy sits beside the replicas' slices in the workgroup's LDS: when one slice with the Kahn scratch in it leaves no room for
            // the copy (about 3,100 - 3,400 nodes without owners plus a large list slice), the copy stays in HBM instead of an over-size launch
            if (gl.prec_static) {
                const int ns2 = ctx->has_scalar_model ? ctx->sm.n : 0;
                const bool tables2 = ctx->has_scalar_model && ctx->sm.tables();
                const GCarve<int16_t> cv2(ns2, ctx->lm.V, ctx->lm.n_cap, gl.has_nearby ? ctx->lm.dim : 0, gl.kopt_nearby, gl.n, gl.has_ruin ? (ctx->lm.leg16 ? 2 : 1) : 0,
                                          ctx->lm.dim, gl.prec.n, tables2 ? ctx->sm.n_values : 0, tables2 && ctx->sm.run_level >= 0 ? ctx->sm.run_P : 0, 0, false);
                if (cv2.total + 1024 + (size_t)gl.prec_static > SF_LDS_BUDGET) gl.prec_static = 0, gl.prec_static_slim = 0;
            }
        }
        // grouped trial evaluator (sf_prec_group.h): T trials per wavefront with private LDS scratch.  SF_AMD_PREC_GROUPS = 0 / 2 / 4 / 8 / 16
        gl.prec_groups = 0;
        if (gl.prec.on && gl.prec_lds && gl.prec_static && !gl.prec_static_slim) {  // (its node records live in the FULL shared static copy)
            // default: as many trials per wave as the graph's width allows -- a Kahn round pops at most one node per list, so lane groups of
            // the largest power of two <= the list count (5 machines: 4 lanes, 16 trials; 10: 8 lanes, 8 trials) -- halved until the scratch
            // fits: under 14 KB, or the replica's precedence state (scratch + 16 B per node of Kahn arrays + ~2.5 KB) under 20 KB, which
            // keeps eight replicas on a CU.  200 nodes: 4; 300: 2; 1,000: off (50 x 20: 34.8 -> 26.0 M moves/s with 2)
            int T = 0;
            {
                int g = 1;
                while (g * 2 <= (ctx->lm.V > 2 ? ctx->lm.V : 2)) g *= 2;
                for (int t = 64 / g > 16 ? 16 : 64 / g; t >= 2; t >>= 1) {
                    const size_t b = pgrp_bytes(gl.prec.n, t, ctx->lm.V);
                    if (b <= 14 * 1024 || b + (size_t)gl.prec.n * 16 + 2560 <= 20 * 1024) {
                        T = t;
                        break;
                    }
                }
            }
            if (const char* e = std::getenv("SF_AMD_PREC_GROUPS")) {
                T = std::atoi(e);
                if (T != 2 && T != 4 && T != 8 && T != 16) T = 0;
                while (T > 1 && pgrp_bytes(gl.prec.n, T, ctx->lm.V) > 40 * 1024) T >>= 1;
            }
            gl.prec_groups = T > 1 ? T : 0;
        }
        gl.prec_sweep = (gl.prec.on && !gl.prec_lds && !gl.prec_inc && std::getenv("SF_AMD_PREC_NO_SWEEP") == nullptr) ? 1 : 0;
        if (gl.plf.on) gl.prec_inc = gl.prec_sweep = 0;  // the critical-path leaf re-evaluates in the main scratch arrays: one full evaluation per trial
        if (gl.prec_sweep && !ctx->pm.elane) {  // [R][n][64] earliest starts of the trials in flight
            int rc = dalloc(ctx, &ctx->pm.elane, (size_t)ctx->R * (size_t)ctx->pm.n * 64);
            if (rc) return rc;
            gl.prec.elane = ctx->pm.elane;
        }
    }
    if (gl.prec.on && gl.has_ruin) {  // ruin leaf on a precedence model (i16 values)
        if (ctx->levels <= 2)
            return launch_mixed_t<2, int16_t, true, true>(ctx, p, gl, grid, trace);
        return launch_mixed_t<4, int16_t, true, true>(ctx, p, gl, grid, trace);
    }
    if (gl.prec.on) {  // ListPrecedenceMakespanConstraint: its own instantiations
        if (ctx->has_scalar_model && ctx->sm.n_values <= 127 && ctx->sm.n >= 1024) {  // one-byte value array (C4: 4 waves per CU instead of 3)
            if (ctx->levels <= 2)
                return launch_mixed_t<2, int8_t, false, true>(ctx, p, gl, grid, trace);
            return launch_mixed_t<4, int8_t, false, true>(ctx, p, gl, grid, trace);
        }
        if (ctx->levels <= 2)
            return launch_mixed_t<2, int16_t, false, true>(ctx, p, gl, grid, trace);
        return launch_mixed_t<4, int16_t, false, true>(ctx, p, gl, grid, trace);
    }
    if (gl.has_ruin) {  // the ruin leaf has its own instantiations (i16 values only)
        if (ctx->levels <= 2)
            return launch_mixed_t<2, int16_t, true>(ctx, p, gl, grid, trace);
        return launch_mixed_t<4, int16_t, true>(ctx, p, gl, grid, trace);
    }
    if (ctx->has_scalar_model && ctx->sm.n_values <= 127 && ctx->sm.n >= 1024) {
        if (ctx->levels <= 2)
            return launch_mixed_t<2, int8_t>(ctx, p, gl, grid, trace);
        return launch_mixed_t<4, int8_t>(ctx, p, gl, grid, trace);
    }
    if (ctx->levels <= 2)
        return launch_mixed_t<2, int16_t>(ctx, p, gl, grid, trace);
    return launch_mixed_t<4, int16_t>(ctx, p, gl, grid, trace);
}

extern "C" {

static int launch_search(sf_ctx* ctx, SearchParams& p, int grid, bool trace) {
    // the 2-leaf nearby union has its own engines; every other union runs in the generic N-leaf engine
    // a configured root union (order / weights other than the default policy's) runs in the generic engine too
    bool nearby_scalar = false;  // the nearby scalar leaves live in the generic engine
    for (auto& s : ctx->selectors) nearby_scalar = nearby_scalar || s.kind == SF_SEL_NEARBY_SCALAR_CHANGE || s.kind == SF_SEL_NEARBY_SCALAR_SWAP;
    if (nearby_scalar || (ctx->has_list_model && ctx->has_scalar_model) || (ctx->has_list_model && has_plain_list_leaves(ctx)) || union_is_custom(ctx) ||
        (ctx->has_list_model && ctx->pm.on))  // the precedence constraint is scored by the generic engine only
        return launch_mixed(ctx, p, grid, trace);
    if (ctx->has_list_model) {
        int rc = fill_list_leaves(ctx, p);
        if (rc) return rc;
        return launch_list_search(ctx, p, grid, trace);
    }
    return launch_scalar_search(ctx, p, grid, trace);
}

static int fold_events(sf_ctx* ctx);

static int32_t solve_launch(sf_ctx* ctx, int64_t n_steps, int64_t move_budget);

int32_t sf_solve_steps(sf_ctx* ctx, int64_t n_steps) { return solve_launch(ctx, n_steps, 0); }

int32_t sf_solve_moves(sf_ctx* ctx, int64_t max_steps, int64_t move_budget) {
    if (move_budget <= 0) return fail(ctx, SF_ERR_INVALID, "sf_solve_moves: move_budget must be positive");
    // the per-launch candidate counter of a replica is 32 bits wide (a launch that long would run for hours)
    if (move_budget >= ((int64_t)1 << 31)) return fail(ctx, SF_ERR_INVALID, "sf_solve_moves: move_budget must be < 2^31 per launch");
    return solve_launch(ctx, max_steps, move_budget);
}

static int32_t solve_launch(sf_ctx* ctx, int64_t n_steps, int64_t move_budget) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || !ctx->search_alloc) return fail(ctx, SF_ERR_INVALID, "sf_phase_start first");
    if (n_steps <= 0) return SF_OK;
    if (n_steps >= ((int64_t)1 << 31)) return fail(ctx, SF_ERR_INVALID, "at most 2^31 - 1 steps per launch");
    if (ctx->events.size() >= 1024) {  // a long-lived context never holds more than 1024 event pairs
        int rcf = fold_events(ctx);
        if (rcf) return rcf;
    }
    SearchParams p = ctx->sp;
    fill_search_params(ctx, p);
    p.n_steps = n_steps;
    p.move_budget = move_budget;
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    int rc = launch_search(ctx, p, ctx->R, false);
    if (rc) {
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        return rc;
    }
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
    ctx->events.push_back({e0, e1});
    return SF_OK;
}

// add the finished launches' durations to the running totals and release their events
static int fold_events(sf_ctx* ctx) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& ev : ctx->events) {
        float ms = 0;
        hipError_t e = hipEventElapsedTime(&ms, ev.first, ev.second);
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
        if (e != hipSuccess) {
            ctx->events.clear();
            return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
        }
        ctx->prof_ms += ms;
        ctx->prof_launches += 1;
    }
    ctx->events.clear();
    return SF_OK;
}

int32_t sf_profile_solve(sf_ctx* ctx, double* out_ms, int64_t* out_launches) {
    DeviceGuard _dev(ctx);
    if (!ctx) return SF_ERR_INVALID;
    int rc = fold_events(ctx);
    if (rc) return rc;
    if (out_ms) *out_ms = ctx->prof_ms;
    if (out_launches) *out_launches = ctx->prof_launches;
    ctx->prof_ms = 0;
    ctx->prof_launches = 0;
    return SF_OK;
}

static int fetch_trace(sf_ctx* ctx, sf_move_t* out_moves, int64_t* out_scores, int32_t* out_flags, int64_t cap,
                       int64_t* out_count) {
    int64_t n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, ctx->d_trace_count, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (out_count) *out_count = n;
    int64_t m = n < cap ? n : cap;
    if (m > ctx->trace_cap) m = ctx->trace_cap;
    if (m > 0) {
        if (out_moves) HIPCHK(ctx, hipMemcpyAsync(out_moves, ctx->d_trace_moves, (size_t)m * 24, hipMemcpyDeviceToHost, ctx->stream));
        if (out_scores) HIPCHK(ctx, hipMemcpyAsync(out_scores, ctx->d_trace_scores, (size_t)m * ctx->levels * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (out_flags) HIPCHK(ctx, hipMemcpyAsync(out_flags, ctx->d_trace_flags, (size_t)m * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n > cap) return fail(ctx, SF_ERR_CAPACITY, "trace capacity too small");
    return SF_OK;
}

int32_t sf_step_generate(sf_ctx* ctx, int32_t replica, uint64_t step_index, uint64_t step_seed, int32_t order,
                         sf_move_t* out_moves, int64_t* out_scores, int32_t* out_doable, int64_t cap,
                         int64_t* out_count) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (replica < 0 || replica >= ctx->R || cap <= 0 || !out_moves) return fail(ctx, SF_ERR_INVALID, "bad sf_step_generate arguments");
    int rc = alloc_search(ctx);
    if (rc) return rc;
    if ((rc = ensure_trace(ctx, cap))) return rc;
    SearchParams p = ctx->sp;
    fill_search_params(ctx, p);
    p.acceptor = 2;  // never accept: pure enumeration + trial scores
    p.forager = 2;
    p.order = order;
    p.dry_run = 1;
    p.dry_step_index = step_index;
    p.dry_step_seed = step_seed;
    p.n_steps = 1;
    p.replica_base = replica;
    p.trace_replica = replica;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_trace_count, 0, 8, ctx->stream));
    if ((rc = launch_search(ctx, p, 1, true))) return rc;
    std::vector<int32_t> flags((size_t)cap);
    rc = fetch_trace(ctx, out_moves, out_scores, flags.data(), cap, out_count);
    if (out_doable && out_count)
        for (int64_t i = 0; i < *out_count && i < cap; ++i) out_doable[i] = flags[(size_t)i] & 1;
    return rc;
}

int32_t sf_solve_step_traced(sf_ctx* ctx, int32_t replica, sf_move_t* out_moves, int64_t* out_scores,
                             int32_t* out_flags, int64_t cap, int64_t* out_count, int32_t* out_applied,
                             sf_move_t* out_applied_move) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || !ctx->search_alloc) return fail(ctx, SF_ERR_INVALID, "sf_phase_start first");
    if (replica < 0 || replica >= ctx->R || cap <= 0) return fail(ctx, SF_ERR_INVALID, "bad arguments");
    int rc = ensure_trace(ctx, cap);
    if (rc) return rc;
    SearchParams p = ctx->sp;
    fill_search_params(ctx, p);
    p.n_steps = 1;
    p.trace_replica = replica;
    HIPCHK(ctx, hipMemsetAsync(ctx->d_trace_count, 0, 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_trace_applied, 0, 32, ctx->stream));
    if ((rc = launch_search(ctx, p, ctx->R, true))) return rc;
    rc = fetch_trace(ctx, out_moves, out_scores, out_flags, cap, out_count);
    int32_t ap[8];
    HIPCHK(ctx, hipMemcpyAsync(ap, ctx->d_trace_applied, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (out_applied) *out_applied = ap[0];
    if (out_applied_move && ap[0]) std::memcpy(out_applied_move, &ap[1], 24);
    return rc;
}

int32_t sf_get_stats(sf_ctx* ctx, int32_t replica, sf_stats* out) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->search_alloc || !out || replica < 0 || replica >= ctx->R) return fail(ctx, SF_ERR_INVALID, "bad sf_get_stats");
    static_assert(sizeof(sf_stats) == SF_STATS_WORDS * 8, "sf_stats layout");
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->sp.stats + (size_t)replica * SF_STATS_WORDS, sizeof(sf_stats), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

int32_t sf_get_stats_sum(sf_ctx* ctx, sf_stats* out) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->search_alloc || !out) return fail(ctx, SF_ERR_INVALID, "bad sf_get_stats_sum");
    std::vector<uint64_t> all((size_t)ctx->R * SF_STATS_WORDS);
    HIPCHK(ctx, hipMemcpyAsync(all.data(), ctx->sp.stats, all.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t sum[SF_STATS_WORDS] = {0};
    for (int r = 0; r < ctx->R; ++r)
        for (int k = 0; k < SF_STATS_WORDS; ++k) sum[k] += all[(size_t)r * SF_STATS_WORDS + k];
    std::memcpy(out, sum, sizeof(sf_stats));
    return SF_OK;
}

int32_t sf_get_best_scores(sf_ctx* ctx, int64_t* out) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized || !out) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    return download_scores(ctx, ctx->has_list_model ? ctx->lm.best_score : ctx->sm.best_score, out);
}

// Elite migration between the replicas of one context (extension; see k_list_migrate).  Replicas are ranked by their best score
// (descending, ties to the lower index); the n_replace last ones adopt the best solution of the n_elite first ones, adopter i
// taking elite i % n_elite.
int32_t sf_portfolio_migrate_local(sf_ctx* ctx, int32_t n_elite, int32_t n_replace, int32_t* out_adopted) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->initialized) return fail(ctx, SF_ERR_INVALID, "sf_initialize first");
    if (!ctx->search_alloc) return fail(ctx, SF_ERR_INVALID, "sf_phase_start first");
    if (!ctx->has_list_model || ctx->has_scalar_model)
        return fail(ctx, SF_ERR_UNSUPPORTED, "elite migration: list-only models (the adopted state is the list class's best solution)");
    if (ctx->cfg.acceptor == SF_ACCEPT_SIMULATED_ANNEALING)
        return fail(ctx, SF_ERR_UNSUPPORTED, "elite migration: HillClimbing / LateAcceptance / DiversifiedLateAcceptance (an annealing replica keeps a temperature schedule)");
    if (n_elite < 1 || n_replace < 0 || (int64_t)n_elite + n_replace > ctx->R)
        return fail(ctx, SF_ERR_INVALID, "elite migration: n_elite >= 1, n_replace >= 0, n_elite + n_replace <= replicas");
    if (out_adopted) *out_adopted = 0;
    if (n_replace == 0) return SF_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> bs((size_t)ctx->R * ctx->levels);
    if (int rc = download_scores(ctx, ctx->lm.best_score, bs.data())) return rc;
    std::vector<int32_t> order(ctx->R), src(ctx->R);
    for (int r = 0; r < ctx->R; ++r) order[r] = src[r] = r;
    const int L = ctx->levels;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        for (int k = 0; k < L; ++k) {
            const int64_t x = bs[(size_t)a * L + k], y = bs[(size_t)b * L + k];
            if (x != y) return x > y;
        }
        return false;
    });
    int adopted = 0;
    for (int i = 0; i < n_replace; ++i) {
        const int32_t adopter = order[(size_t)ctx->R - 1 - i], elite = order[(size_t)(i % n_elite)];
        bool same = true;  // an adopter that already holds the elite's score keeps searching from its own state
        for (int k = 0; k < L; ++k) same = same && bs[(size_t)adopter * L + k] == bs[(size_t)elite * L + k];
        if (same) continue;
        src[adopter] = elite;
        adopted += 1;
    }
    if (adopted) {
        int32_t* d_src = nullptr;
        hipError_t e = hipMalloc((void**)&d_src, (size_t)ctx->R * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_src, src.data(), (size_t)ctx->R * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_list_migrate, dim3(ctx->R), dim3(256), 0, ctx->stream, ctx->lm, ctx->sp, d_src);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_src);
        if (e != hipSuccess) return fail(ctx, SF_ERR_HIP, hipGetErrorString(e));
    }
    if (out_adopted) *out_adopted = adopted;
    return SF_OK;
}

int32_t sf_download_list(sf_ctx* ctx, int32_t replica, int32_t d, uint32_t* out_off, uint32_t* out_vals, int32_t best) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->has_list_model || d != ctx->list_desc || replica < 0 || replica >= ctx->R)
        return fail(ctx, SF_ERR_INVALID, "bad sf_download_list");
    const ListModel& m = ctx->lm;
    const uint32_t* so = (best ? m.best_off : m.off) + (size_t)replica * (m.V + 1);
    const uint32_t* sv = (best ? m.best_visits : m.visits) + (size_t)replica * m.n_cap;
    HIPCHK(ctx, hipMemcpyAsync(out_off, so, (size_t)(m.V + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t total = out_off[m.V];
    if (total) HIPCHK(ctx, hipMemcpyAsync(out_vals, sv, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

int32_t sf_download_scalar(sf_ctx* ctx, int32_t replica, int32_t d, int32_t var, int32_t* out, int32_t best) {
    DeviceGuard _dev(ctx);
    if (!ctx || !ctx->has_scalar_model || d != ctx->scalar_desc || replica < 0 || replica >= ctx->R)
        return fail(ctx, SF_ERR_INVALID, "bad sf_download_scalar");
    (void)var;
    const ScalarModel& m = ctx->sm;
    const int32_t* src = (best ? m.best_vals : m.vals) + (size_t)replica * m.n;
    HIPCHK(ctx, hipMemcpyAsync(out, src, (size_t)m.n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return SF_OK;
}

}  // extern "C"

#ifdef SF_PHASE_PROFILE
extern "C" int32_t sf_debug_ruin_phases(uint64_t* out8) {  // diagnostic builds only: shader clocks inside ruin_recreate
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_rphase), 64) != hipSuccess) return SF_ERR_HIP;
    uint64_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sf::g_rphase), z, 64);
    return SF_OK;
}
extern "C" int32_t sf_debug_phases(uint64_t* out8) {  // diagnostic builds only (scripts/phase_probe.py)
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(sf::g_phase), 64) != hipSuccess) return SF_ERR_HIP;
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(sf::g_phase), z, 64);
    return SF_OK;
}
#endif

#include "sf_api_scalar.inc"
#include "sf_portfolio.inc"
#include "sf_candidate_trace.inc"
