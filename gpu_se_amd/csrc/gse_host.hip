// gse_host.hip -- host-side plumbing of the C ABI: error text, argument checks, kernel-argument packing,
// workspace layout, gse_version / gse_last_error / gse_workspace_bytes.
#include "gse_host.h"

#include <math.h>
#include <chrono>
#include <string.h>

namespace gse {

char* last_error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int check_pf_geometry(const char* fn, int B, int64_t N, int64_t ld) {
    GSE_REQUIRE(B >= 1 && B <= GSE_MAX_B, "%s: B = %d is outside 1..%d", fn, B, GSE_MAX_B);
    GSE_REQUIRE(N >= 1, "%s: N = %lld must be positive", fn, (long long)N);
    GSE_REQUIRE(ld >= N && (ld & 3) == 0, "%s: ld = %lld must be >= N = %lld and a multiple of 4", fn, (long long)ld,
                (long long)N);
    GSE_REQUIRE(tiles_of(N) <= 0x7fffffff / 2, "%s: N = %lld is too large", fn, (long long)N);
    return GSE_OK;
}

int check_mixture(const char* fn, const gse_mixture* m, int nx) {
    GSE_REQUIRE(m->nd >= 1 && m->nd <= GSE_MAX_ND, "%s: mixture has nd = %d, supported 1..%d", fn, m->nd, GSE_MAX_ND);
    GSE_REQUIRE(m->nx == nx, "%s: mixture has nx = %d, this entry point needs nx = %d", fn, m->nx, nx);
    return GSE_OK;
}

void pack_io(FilterIOs& io, int B, const double* u, const double* z) {
    for (int b = 0; b < B; ++b) {
        io.f[b].u0 = u ? u[2 * b + 0] : 0.0;
        io.f[b].u1 = u ? u[2 * b + 1] : 0.0;
        io.f[b].z0 = z ? z[2 * b + 0] : 0.0;
        io.f[b].z1 = z ? z[2 * b + 1] : 0.0;
    }
}

void pack_draw(DrawParams& dp, const gse_mixture& m) {
    memset(&dp, 0, sizeof(dp));
    dp.nd = m.nd;
    dp.nx = m.nx;
    double total = 0.0;
    for (int d = 0; d < m.nd; ++d) total += (double)m.weights[d];
    double run = 0.0;
    dp.diagonal = 1;
    for (int d = 0; d < m.nd; ++d) {
        run += (double)m.weights[d];
        dp.cumw[d] = d == m.nd - 1 ? 1.0f : (float)(run / total);      // (the last one IS 1: mixture_transform5 relies on it)
        int at = 0;
        for (int k = 0; k < GSE_NX; ++k) {
            dp.mean[d][k] = (k < m.nx ? m.means[d][k] : 0.f) + 0.0f;       // (-0 -> +0: mixture_transform5)
            for (int j = 0; j <= k; ++j) {
                const float l = (k < m.nx && j < m.nx) ? m.chol[d][k][j] : 0.f;
                if (j < k && l != 0.f) dp.diagonal = 0;
                dp.chol[d][at++] = l;
            }
        }
    }
    for (int d = m.nd; d < GSE_MAX_ND; ++d) dp.cumw[d] = 2.0f;
}

void pack_pdf2(Pdf2& p, const gse_mixture& m) {
    memset(&p, 0, sizeof(p));
    p.nd = m.nd;
    for (int d = 0; d < m.nd; ++d) {
        p.cw[d] = (double)(m.consts[d] * m.weights[d]);          // fp32 product, then widened (MultivariateGaussianSum.py:61)
        p.l2cw[d] = p.cw[d] > 0.0 ? log2(p.cw[d]) : -INFINITY;
        for (int i = 0; i < 2; ++i) {
            p.mean[d][i] = (double)m.means[d][i];
            for (int j = 0; j < 2; ++j) p.inv[d][i][j] = m.inv_cov[d][i][j];
        }
    }
}

static int64_t round256(int64_t v) { return (v + 255) & ~(int64_t)255; }

Workspace carve_workspace(void* base, int B, int64_t N) {
    const int64_t T = tiles_of(N);
    Workspace ws;
    char* p = static_cast<char*>(base);
    int64_t at = 0;
    ws.tile_sums = reinterpret_cast<double*>(p + at);
    at += round256((int64_t)B * T * 8);
    ws.tile_prefix = reinterpret_cast<double*>(p + at);
    at += round256((int64_t)B * (T + 1) * 8);
    ws.partials = reinterpret_cast<double*>(p + at);
    at += round256((int64_t)B * kMomentBlocks * kMomentWidth * 8);
    ws.totals = reinterpret_cast<double*>(p + at);
    at += round256((int64_t)B * 8);
    for (int k = 0; k < 2; ++k) {
        ws.scan_sums[k] = reinterpret_cast<double*>(p + at);
        at += round256((int64_t)B * T * 8);
        ws.scan[k] = reinterpret_cast<double*>(p + at);
        at += round256((int64_t)B * T * kBlock * 8);
    }
    ws.cumsum = reinterpret_cast<double*>(p + at);
    at += round256((int64_t)B * N * 8);
    ws.wscale = reinterpret_cast<double*>(p + at);
    at += round256((int64_t)B * 8);
    for (int k = 0; k < 2; ++k) {
        ws.qsums[k] = reinterpret_cast<double*>(p + at);
        at += round256((int64_t)B * T * 4 * 8);
    }
    for (int k = 0; k < 2; ++k) {
        ws.samples[k] = reinterpret_cast<float*>(p + at);
        at += round256((int64_t)B * 64 * 8 * 4);
    }
    ws.kexact = reinterpret_cast<uint32_t*>(p + at);
    ws.wstatus = ws.kexact + GSE_MAX_B;                  // (the same 256 bytes: 2 x 32 words)
    static_assert(2 * GSE_MAX_B * 4 <= 256, "kexact and wstatus share one 256-byte block");
    at += round256((int64_t)B * 4);
    ws.counters = reinterpret_cast<uint32_t*>(p + at);
    at += round256((int64_t)B * 4);
    ws.bytes = at;
    return ws;
}

}  // namespace gse

extern "C" {

int gse_version(void) { return GSE_VERSION; }

const char* gse_last_error(void) { return gse::last_error_buffer(); }

int gse_stream_synchronize(void* stream, int spin) {
    hipStream_t st = (hipStream_t)stream;
    // Busy-wait on the stream's state: the caller's next launch should follow the end of this work by a microsecond, not by
    // the wake-up latency of a blocking wait (the reference's benchmark protocol times every method call by itself,
    // results/pf_openloop/pf_run_seq.py:15-130 -- there the wait is most of what a small filter costs).  Bounded: after
    // 20 ms of polling the work is no short kernel any more (or hangs) and the blocking wait takes over -- a host core is not
    // spun for ever inside a library call.
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int polls = 0;; ++polls) {
            const hipError_t e = hipStreamQuery(st);
            if (e == hipSuccess) return GSE_OK;
            if (e != hipErrorNotReady) return gse::fail(GSE_ERR_HIP, "gse_stream_synchronize: %s", hipGetErrorString(e));
            if ((polls & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    const hipError_t e = hipStreamSynchronize(st);
    return e == hipSuccess ? GSE_OK : gse::fail(GSE_ERR_HIP, "gse_stream_synchronize: %s", hipGetErrorString(e));
}

int64_t gse_workspace_status_offset(int B, int64_t N) {
    if (B < 1 || B > GSE_MAX_B || N < 1)
        return gse::fail(GSE_ERR_ARG, "gse_workspace_status_offset: bad B = %d or N = %lld", B, (long long)N);
    gse::Workspace ws = gse::carve_workspace(nullptr, B, N);
    return (int64_t)(reinterpret_cast<char*>(ws.wstatus) - static_cast<char*>(nullptr));
}

int64_t gse_workspace_bytes(int B, int64_t N) {
    if (B < 1 || B > GSE_MAX_B || N < 1) return gse::fail(GSE_ERR_ARG, "gse_workspace_bytes: bad B = %d or N = %lld", B, (long long)N);
    return gse::carve_workspace(nullptr, B, N).bytes;
}

}  // extern "C"
