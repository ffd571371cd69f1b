// gse_pf_loop.hip -- the resident closed loop of the particle filter driven from C++ (round 4).
//
// gpu_se_amd/sim_base.py BatchedSimulation(resident=True) keeps everything of a tick on the device -- three launches per
// control tick -- but enqueued them from Python: 107 us of host time per tick whatever B and N, three times the device time of
// one real-time loop (B = 1, N <= 2^18).  This entry point is that driver's tick loop and the deferred-stage bookkeeping of
// filter.ParallelParticleFilter (resample and predict ride the next update's launch) restated in C++ on top of the SAME entry
// points of this library (gse_pf_step, gse_pf_moments_resampled_tail, gse_plant_step, gse_pf_moments): the same launches
// with the same arguments in the same order, so the records are those of the Python loop bit for bit -- at a few microseconds
// of host time per launch.  Reference: the loop of sim_base.py:243-309, for B rollouts.
#include "gse_host.h"

#include <vector>

using namespace gse;

namespace {

struct PfLoopState {
    const gse_pf_loop_spec* s;
    const double *u_cur = nullptr, *z_cur = nullptr;    // this launch's inputs / measurements; io_dev: device pointers [B][2]
    bool io_dev = true;
    float *x, *x_alt, *w, *w_alt;
    bool pend_gather = false, pend_predict = false;
    int gather_slot = -1, scan_slot = -1;
    bool samples_ok = false;                            // GSE_OP_SAMPLES: the slot's sample rows describe the particles
    bool w_uniform = false;                             // every weight is 1 / N (a resample's copy ran and no update since):
                                                        // an update without a copy in its launch may pass GSE_OP_UNIFORM_WEIGHTS
    gse_rng rng;
    const double* r_cur;
    hipStream_t st;

    // filter/particle.py _launch_step for inputs that live on the device, Philox noise, the built-in model
    int launch_step(bool update) {
        const bool cluster = (s->flags & GSE_PF_LOOP_CLUSTER) != 0;
        if (cluster) return launch_cluster(update);
        const bool whole = pend_gather && gather_slot >= 0 && pend_predict && update;
        if (io_dev && pend_gather && !whole && (pend_predict || update)) {
            // device-resident u / z are read by the single-launch cycle or by the variants without gather: copy the ancestors first
            const bool keep = pend_predict;
            pend_predict = false;
            if (int rc = launch_step(false)) return rc;
            pend_predict = keep;
        }
        uint32_t ops = ((s->flags & GSE_PF_LOOP_WEIGHT_SCALE) && (update || pend_gather)) ? GSE_OP_WEIGHT_SCALE : 0u;
        const bool gather = pend_gather;
        if (ops && update && !gather && w_uniform) ops |= GSE_OP_UNIFORM_WEIGHTS;      // (filter/particle.py _launch_step)
        const bool swap_x = gather || (ops && update && pend_predict);
        float* x_out = swap_x ? x_alt : x;
        const bool swap_w = gather || (ops && update);
        float* w_out = swap_w ? w_alt : w;
        if (gather) {
            ops |= GSE_OP_GATHER;
            if (gather_slot >= 0) ops |= GSE_OP_SCAN_IN | (gather_slot ? GSE_OP_SCAN_IN_SLOT1 : 0u) | (samples_ok ? GSE_OP_SAMPLES : 0u);
        }
        if (pend_predict) ops |= GSE_OP_PREDICT | (io_dev ? GSE_OP_U_DEVICE : 0u);
        int scan_out = -1;
        if (update) {
            scan_out = (gather && gather_slot == 0) ? 1 : 0;
            ops |= GSE_OP_UPDATE | (io_dev ? GSE_OP_Z_DEVICE : 0u) | GSE_OP_SCAN_OUT | (scan_out ? GSE_OP_SCAN_OUT_SLOT1 : 0u);
        }
        if (int rc = gse_pf_step(ops, x, x_out, w, w_out, nullptr, s->ld, s->B, s->N, u_cur, s->dt, z_cur, r_cur, s->state_pdf, &rng,
                                 nullptr, s->measurement_pdf, s->island, s->workspace, st))
            return rc;
        if (pend_predict) rng.step += 1;
        if (swap_x) std::swap(x, x_alt);
        if (swap_w) std::swap(w, w_alt);
        if (update) w_uniform = false; else if (gather) w_uniform = true;
        samples_ok = update && (ops & GSE_OP_WEIGHT_SCALE);      // (nothing but these launches touches the particles while the loop runs)
        pend_gather = false;
        gather_slot = -1;
        pend_predict = false;
        if (update) scan_slot = scan_out;
        return GSE_OK;
    }
    // filter/particle.py _launch_cluster: four workgroups per tile (GSE_OP_CLUSTER), any stage combination in one launch
    int launch_cluster(bool update) {
        const bool gather = pend_gather;
        uint32_t ops = GSE_OP_CLUSTER | (((s->flags & GSE_PF_LOOP_WEIGHT_SCALE) && (update || gather)) ? GSE_OP_WEIGHT_SCALE : 0u);
        const bool swap_x = gather || ((ops & GSE_OP_WEIGHT_SCALE) && update && pend_predict);
        float* x_out = swap_x ? x_alt : x;
        const bool swap_w = gather || ((ops & GSE_OP_WEIGHT_SCALE) && update);
        float* w_out = swap_w ? w_alt : w;
        if ((ops & GSE_OP_WEIGHT_SCALE) && update && !gather && w_uniform) ops |= GSE_OP_UNIFORM_WEIGHTS;
        if (gather) ops |= GSE_OP_GATHER | GSE_OP_SCAN_IN | (gather_slot ? GSE_OP_SCAN_IN_SLOT1 : 0u);
        if (pend_predict) ops |= GSE_OP_PREDICT | (io_dev ? GSE_OP_U_DEVICE : 0u);
        int scan_out = -1;
        if (update) {
            scan_out = (gather && gather_slot == 0) ? 1 : 0;
            ops |= GSE_OP_UPDATE | (io_dev ? GSE_OP_Z_DEVICE : 0u) | GSE_OP_SCAN_OUT | (scan_out ? GSE_OP_SCAN_OUT_SLOT1 : 0u);
        }
        if (int rc = gse_pf_step(ops, x, x_out, w, w_out, nullptr, s->ld, s->B, s->N, u_cur, s->dt, z_cur, r_cur, s->state_pdf, &rng,
                                 nullptr, s->measurement_pdf, s->island, s->workspace, st))
            return rc;
        if (pend_predict) rng.step += 1;
        if (swap_x) std::swap(x, x_alt);
        if (swap_w) std::swap(w, w_alt);
        if (update) w_uniform = false; else if (gather) w_uniform = true;
        pend_gather = false;
        gather_slot = -1;
        pend_predict = false;
        if (update) scan_out_slot(scan_out);
        return GSE_OK;
    }
    void scan_out_slot(int slot) { scan_slot = slot; }
    int flush() { return (pend_gather || pend_predict) ? launch_step(false) : GSE_OK; }
};

}  // namespace

extern "C" int gse_pf_closed_loop(gse_pf_loop_spec* s, void* stream) {
    GSE_REQUIRE(s, "gse_pf_closed_loop: null spec");
    GSE_REQUIRE(s->B >= 1 && s->B <= GSE_MAX_B && s->ticks >= 2, "gse_pf_closed_loop: bad B / ticks");
    GSE_REQUIRE(s->N >= 1 && s->N <= ((int64_t)1 << 21), "gse_pf_closed_loop: 1 <= N <= 2^21 (the single-launch cycle's range)");
    GSE_REQUIRE(!(s->flags & GSE_PF_LOOP_CLUSTER) || s->N <= ((int64_t)256 * GSE_TILE),
                "gse_pf_closed_loop: GSE_PF_LOOP_CLUSTER takes at most 256 tiles (N <= 2^18)");
    GSE_REQUIRE(s->x[0] && s->x[1] && s->w[0] && s->w[1] && s->workspace && s->state_pdf && s->measurement_pdf,
                "gse_pf_closed_loop: null filter argument");
    GSE_REQUIRE(s->measurement_pdf->nd == 2, "gse_pf_closed_loop: the single-launch cycle needs a 2-component measurement mixture");
    GSE_REQUIRE(s->law && s->glue && s->X && s->u && s->z && s->y_pred && s->known && s->plant_state_noise && s->plant_meas_noise,
                "gse_pf_closed_loop: null plant / controller argument");
    GSE_REQUIRE(s->schedule && s->us && s->xs && s->ys && s->ys_meas && s->moments && s->status, "gse_pf_closed_loop: null record");
    const int B = s->B;
    int n_control = 0;
    for (int t = 1; t < s->ticks; ++t) n_control += (s->schedule[t] & 2) ? 1 : 0;
    GSE_REQUIRE(n_control == 0 || s->r, "gse_pf_closed_loop: the resamples need their uniform numbers r [updates][B]");
    PfLoopState f;
    f.s = s;
    f.x = s->x[0]; f.x_alt = s->x[1]; f.w = s->w[0]; f.w_alt = s->w[1];
    f.rng = s->rng;
    f.r_cur = s->r;
    f.u_cur = s->u;
    f.z_cur = s->z;
    f.st = (hipStream_t)stream;
    f.w_uniform = s->w_uniform != 0;
    gse_rng plant_rng = s->plant_rng;
    gse_loop_tail tail = {};
    tail.law = s->law; tail.glue = s->glue; tail.u = s->u; tail.y_pred = s->y_pred; tail.known = s->known; tail.X = s->X;
    tail.dt = s->dt; tail.state_noise = s->plant_state_noise; tail.meas_noise = s->plant_meas_noise; tail.rng = &plant_rng;
    tail.z = s->z;
    const int64_t no = s->glue->n_output_full, ni = s->glue->n_input_full;
    GSE_REQUIRE(no == 5 && ni == 2, "gse_pf_closed_loop: the records are laid out for the bioreactor (5 outputs, 2 inputs)");
    int k_control = 0;
    for (int t = 1; t < s->ticks; ++t) {
        const unsigned char due = s->schedule[t];
        bool control_done = false;
        if (due & 1) {                                     // predict(u, dt): deferred (two in a row: the first one now)
            if (f.pend_predict)
                if (int rc = f.flush()) return rc;
            f.pend_predict = true;
        }
        if (due & 2) {
            if (int rc = f.launch_step(true)) return rc;   // update(u, z): gather + predict + update in one launch
            // resample(r): the update left the tile sums and per-thread prefixes of these weights -- nothing to launch
            f.r_cur = s->r + (int64_t)k_control * B;
            ++k_control;
            GSE_REQUIRE(f.scan_slot >= 0, "gse_pf_closed_loop: internal: no scan slot after an update");
            f.gather_slot = f.scan_slot;
            f.scan_slot = -1;
            f.pend_gather = true;
            // moments of the ensemble the pending resample will produce, their reduction, the controller, the plant
            plant_rng.step = (uint64_t)t;
            tail.y_meas = s->ys_meas + (int64_t)(t - 1) * B * 5;
            tail.status = s->status + (int64_t)t * B;
            tail.y = s->ys + (int64_t)t * B * 5;
            tail.y_meas_out = s->ys_meas + (int64_t)t * B * 5;
            tail.log_x = s->xs + (int64_t)t * B * 5;
            tail.log_u = s->us + (int64_t)t * B * 2;
            const int slot_arg = f.gather_slot | ((s->flags & GSE_PF_LOOP_CLUSTER) ? GSE_SLOT_QUARTERS : 0);
            if (int rc = gse_pf_moments_resampled_tail(f.x, f.w, s->ld, B, s->N, f.r_cur, slot_arg,
                                                       s->moments + (int64_t)t * B * 26, s->workspace, &tail, stream))
                return rc;
            control_done = true;
        }
        if (!control_done) {
            plant_rng.step = (uint64_t)t;
            if (int rc = gse_plant_step(s->X, s->u, s->dt, s->plant_state_noise, s->plant_meas_noise, &plant_rng,
                                        s->ys + (int64_t)t * B * 5, s->ys_meas + (int64_t)t * B * 5, s->z,
                                        s->xs + (int64_t)t * B * 5, s->us + (int64_t)t * B * 2, B, stream))
                return rc;
            if (int rc = f.flush()) return rc;             // this tick's record: the moments of the current ensemble
            if (int rc = gse_pf_moments(f.x, f.w, s->ld, B, s->N, s->moments + (int64_t)t * B * 26, s->workspace, stream)) return rc;
        }
    }
    // what the caller's filter object has to know to carry on
    s->x_current = (f.x == s->x[0]) ? 0 : 1;
    s->w_current = (f.w == s->w[0]) ? 0 : 1;
    s->pend_gather = f.pend_gather ? 1 : 0;
    s->pend_predict = f.pend_predict ? 1 : 0;
    s->gather_slot = f.gather_slot;
    s->scan_slot = f.scan_slot;
    s->samples_ok = f.samples_ok ? 1u : 0u;
    s->w_uniform = f.w_uniform ? 1 : 0;
    s->rng.step = f.rng.step;
    if (f.pend_gather)
        for (int b = 0; b < B; ++b) s->r_pending[b] = f.r_cur[b];
    return GSE_OK;
}

// K open-loop cycles predict(u_k, dt) -> update(u_k, z_k) -> resample(r_k) of B filters, enqueued from C++: the class's deferred
// stages restated as above (one launch per cycle: the pending resample's copy + predict + update), inputs and measurements HOST
// arrays.  What results/pf_openloop/pf_run_seq.py loops over from Python, 11 us of host time per cycle -- more than the device
// needs for a mid-size filter.  The same launches as K rounds of the three class methods: the same particles and weights.
extern "C" int gse_pf_cycles(gse_pf_loop_spec* s, int K, const double* u /* host [K][B][2] */, const double* z /* host [K][B][2] */,
                             void* stream) {
    GSE_REQUIRE(s && u && z && K >= 1, "gse_pf_cycles: bad argument");
    GSE_REQUIRE(s->B >= 1 && s->B <= GSE_MAX_B && s->N >= 1 && s->N <= ((int64_t)1 << 21), "gse_pf_cycles: bad B / N (N <= 2^21)");
    GSE_REQUIRE(s->x[0] && s->x[1] && s->w[0] && s->w[1] && s->workspace && s->state_pdf && s->measurement_pdf && s->r,
                "gse_pf_cycles: null filter argument");
    GSE_REQUIRE(s->measurement_pdf->nd == 2, "gse_pf_cycles: the single-launch cycle needs a 2-component measurement mixture");
    GSE_REQUIRE(!(s->flags & GSE_PF_LOOP_CLUSTER) || s->N <= ((int64_t)256 * GSE_TILE),
                "gse_pf_cycles: GSE_PF_LOOP_CLUSTER takes at most 256 tiles (N <= 2^18)");
    const int B = s->B;
    PfLoopState f;
    f.s = s;
    f.io_dev = false;
    f.x = s->x[0]; f.x_alt = s->x[1]; f.w = s->w[0]; f.w_alt = s->w[1];
    f.rng = s->rng;
    f.r_cur = s->r;
    f.st = (hipStream_t)stream;
    f.pend_gather = s->pend_gather != 0;                    // (a resample the caller left pending rides the first launch)
    f.gather_slot = s->gather_slot;
    f.samples_ok = f.pend_gather && s->samples_ok != 0;
    f.w_uniform = s->w_uniform != 0;
    if (f.pend_gather) f.r_cur = s->r_pending;
    for (int k = 0; k < K; ++k) {
        f.u_cur = u + (int64_t)k * B * 2;
        f.z_cur = z + (int64_t)k * B * 2;
        f.pend_predict = true;
        if (int rc = f.launch_step(true)) return rc;
        GSE_REQUIRE(f.scan_slot >= 0, "gse_pf_cycles: internal: no scan slot after an update");
        f.r_cur = s->r + (int64_t)k * B;
        f.gather_slot = f.scan_slot;
        f.scan_slot = -1;
        f.pend_gather = true;
    }
    s->x_current = (f.x == s->x[0]) ? 0 : 1;
    s->w_current = (f.w == s->w[0]) ? 0 : 1;
    s->pend_gather = 1;
    s->pend_predict = 0;
    s->gather_slot = f.gather_slot;
    s->scan_slot = -1;
    s->samples_ok = f.samples_ok ? 1u : 0u;
    s->w_uniform = f.w_uniform ? 1 : 0;
    s->rng.step = f.rng.step;
    for (int b = 0; b < B; ++b) s->r_pending[b] = f.r_cur[b];
    return GSE_OK;
}
