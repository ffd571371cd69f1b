// gse_rtc.hip -- run-time compilation of the particle-filter kernels for a user model (see gse_rtc.h).
//
// hiprtc is opened with dlopen (the library named by the caller: torch ships one next to the HIP runtime it loads, ROCm
// another) so that libgse_hip.so itself does not depend on it; everything else goes through the HIP runtime this library
// is already bound to.
#include "gse_rtc.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace gse {

namespace {

// the handful of hiprtc entry points used (hiprtc.h is not included: its types reduce to these)
using rtc_program = void*;
struct Rtc {
    void* dl = nullptr;
    int (*create)(rtc_program*, const char*, const char*, int, const char**, const char**) = nullptr;
    int (*add_name)(rtc_program, const char*) = nullptr;
    int (*compile)(rtc_program, int, const char**) = nullptr;
    int (*log_size)(rtc_program, size_t*) = nullptr;
    int (*log)(rtc_program, char*) = nullptr;
    int (*lowered)(rtc_program, const char*, const char**) = nullptr;
    int (*code_size)(rtc_program, size_t*) = nullptr;
    int (*code)(rtc_program, char*) = nullptr;
    int (*destroy)(rtc_program*) = nullptr;
    const char* (*error_string)(int) = nullptr;
};
Rtc g_rtc;
std::mutex g_mutex;

int open_rtc(const char* path) {
    if (g_rtc.dl) return GSE_OK;
    const char* candidates[] = {path, "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    for (const char* c : candidates) {
        if (!c || !*c) continue;
        g_rtc.dl = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (g_rtc.dl) break;
    }
    if (!g_rtc.dl) return fail(GSE_ERR_UNSUPPORTED, "gse_model_create: cannot open libhiprtc (%s)", dlerror());
#define GSE_RTC_SYM(field, name)                                                                    \
    g_rtc.field = reinterpret_cast<decltype(g_rtc.field)>(dlsym(g_rtc.dl, name));                   \
    if (!g_rtc.field) return fail(GSE_ERR_UNSUPPORTED, "gse_model_create: libhiprtc lacks %s", name)
    GSE_RTC_SYM(create, "hiprtcCreateProgram");
    GSE_RTC_SYM(add_name, "hiprtcAddNameExpression");
    GSE_RTC_SYM(compile, "hiprtcCompileProgram");
    GSE_RTC_SYM(log_size, "hiprtcGetProgramLogSize");
    GSE_RTC_SYM(log, "hiprtcGetProgramLog");
    GSE_RTC_SYM(lowered, "hiprtcGetLoweredName");
    GSE_RTC_SYM(code_size, "hiprtcGetCodeSize");
    GSE_RTC_SYM(code, "hiprtcGetCode");
    GSE_RTC_SYM(destroy, "hiprtcDestroyProgram");
    GSE_RTC_SYM(error_string, "hiprtcGetErrorString");
#undef GSE_RTC_SYM
    return GSE_OK;
}

}  // namespace

struct UserModel {
    std::string header;                 // the generated model (user_increment / user_output, GSE_USER_NX, GSE_USER_NU)
    std::string include_csrc, include_abi;
    int nx = 0, nu = 0, ny = 2;
    bool generic = false;               // a model beyond the fast path's bounds: its kernels are gse_generic_kernels.h
    uint64_t tag = 0;
    std::map<std::string, hipFunction_t> functions;
    std::vector<hipModule_t> modules;
    std::string last_log;
};

int user_model_nx(const UserModel* model) { return model->nx; }

// hiprtc: source -> code object (+ the mangled name of the instance).  No device needed.
static int compile_instance(UserModel* m, const char* name_expression, std::vector<char>& code, std::string& lowered_name) {
    // one translation unit per kernel instance: the model header (registered as an in-memory include), then the kernels
    // (the GS-UKF kernels are a translation unit of their own: a particle-filter instance does not pay for parsing them)
    const bool gsukf = strstr(name_expression, "gse::gsukf_") != nullptr;
    const std::string source = m->generic
        ? std::string("#define GSE_GENERIC_MODEL \"gse_user_model.h\"\n#include \"gse_generic_kernels.h\"\n")
        : std::string("#define GSE_USER_MODEL \"gse_user_model.h\"\n") +
              (gsukf ? "#include \"gse_gsukf_kernels.h\"\n" : "#include \"gse_pf_kernels.h\"\n#include \"gse_rows_kernels.h\"\n");
    const char* header_names[] = {"gse_user_model.h"};
    const char* header_texts[] = {m->header.c_str()};
    rtc_program prog = nullptr;
    int rc = g_rtc.create(&prog, source.c_str(), "gse_user_kernels.hip", 1, header_texts, header_names);
    if (rc) return fail(GSE_ERR_HIP, "hiprtcCreateProgram: %s", g_rtc.error_string(rc));
    g_rtc.add_name(prog, name_expression);
    const std::string i0 = "-I" + m->include_csrc, i1 = "-I" + m->include_abi;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-variable",
                          i0.c_str(), i1.c_str()};
    rc = g_rtc.compile(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    size_t ls = 0;
    g_rtc.log_size(prog, &ls);
    m->last_log.assign(ls + 1, '\0');
    if (ls) g_rtc.log(prog, &m->last_log[0]);
    if (rc) {
        fail(GSE_ERR_HIP, "run-time compilation of %s failed (%s):\n%.1500s", name_expression, g_rtc.error_string(rc),
             m->last_log.c_str());
        g_rtc.destroy(&prog);
        return GSE_ERR_HIP;
    }
    const char* lowered = nullptr;
    size_t cs = 0;
    if (g_rtc.lowered(prog, name_expression, &lowered) || g_rtc.code_size(prog, &cs) || !lowered || !cs) {
        g_rtc.destroy(&prog);
        return fail(GSE_ERR_HIP, "run-time compilation of %s produced no code", name_expression);
    }
    code.resize(cs);
    g_rtc.code(prog, code.data());
    lowered_name = lowered;
    g_rtc.destroy(&prog);
    return GSE_OK;
}

hipFunction_t user_kernel(UserModel* m, const char* name_expression) {
    std::lock_guard<std::mutex> lock(g_mutex);
    // a module is loaded into the CURRENT device: the handles are kept per (device, instance) -- a process that drives a
    // model on a second GPU gets a module of its own there instead of a function of the first device's
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::string key = std::to_string(dev) + ":" + name_expression;
    auto it = m->functions.find(key);
    if (it != m->functions.end()) return it->second;
    std::vector<char> code;
    std::string lowered_name;
    if (compile_instance(m, name_expression, code, lowered_name)) return nullptr;
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleLoadData(&mod, code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&fn, mod, lowered_name.c_str());
    if (e != hipSuccess) {
        fail(GSE_ERR_HIP, "loading the compiled %s failed: %s", name_expression, hipGetErrorString(e));
        if (mod) (void)hipModuleUnload(mod);
        return nullptr;
    }
    m->modules.push_back(mod);
    m->functions[key] = fn;
    return fn;
}

int launch_user_kernel(UserModel* model, const char* name_expression, dim3 grid, dim3 block, size_t dynamic_lds,
                       hipStream_t stream, void** params) {
    hipFunction_t fn = user_kernel(model, name_expression);
    if (!fn) return GSE_ERR_HIP;
    const hipError_t e = hipModuleLaunchKernel(fn, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)dynamic_lds,
                                               stream, params, nullptr);
    if (e != hipSuccess) return fail(GSE_ERR_HIP, "launch of %s failed: %s", name_expression, hipGetErrorString(e));
    return GSE_OK;
}

}  // namespace gse

using namespace gse;

extern "C" {

int gse_model_create(const char* model_header, int nx, int nu, uint64_t tag, const char* include_csrc, const char* include_abi,
                     const char* hiprtc_path, void** model_out) {
    GSE_REQUIRE(model_header && include_csrc && include_abi && model_out, "gse_model_create: null argument");
    GSE_REQUIRE(nx >= 1 && nx <= GSE_NX && nu >= 1 && nu <= GSE_NU, "gse_model_create: nx = %d, nu = %d out of range", nx, nu);
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (int rc = open_rtc(hiprtc_path)) return rc;
    }
    UserModel* m = new UserModel;
    m->header = model_header;
    m->include_csrc = include_csrc;
    m->include_abi = include_abi;
    m->nx = nx;
    m->nu = nu;
    m->tag = tag;
    *model_out = m;
    return GSE_OK;
}

// A model beyond the fast path's bounds (up to 8 states, 4 inputs, 4 outputs): its kernels are those of gse_generic_kernels.h,
// compiled with the model's dimensions as constants and launched by name through gse_model_launch.
int gse_generic_model_create(const char* model_header, int nx, int nu, int ny, uint64_t tag, const char* include_csrc,
                             const char* include_abi, const char* hiprtc_path, void** model_out) {
    GSE_REQUIRE(model_header && include_csrc && include_abi && model_out, "gse_generic_model_create: null argument");
    GSE_REQUIRE(nx >= 1 && nx <= 8 && nu >= 1 && nu <= 4 && ny >= 1 && ny <= 4,
                "gse_generic_model_create: nx = %d, nu = %d, ny = %d out of range (8, 4, 4)", nx, nu, ny);
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        if (int rc = open_rtc(hiprtc_path)) return rc;
    }
    UserModel* m = new UserModel;
    m->header = model_header;
    m->include_csrc = include_csrc;
    m->include_abi = include_abi;
    m->nx = nx; m->nu = nu; m->ny = ny;
    m->generic = true;
    m->tag = tag;
    *model_out = m;
    return GSE_OK;
}

// Launch a kernel of the model's translation unit by its C++ name (compiled on first use): params = one pointer per kernel
// argument, as hipModuleLaunchKernel takes them.
int gse_model_launch(void* model, const char* name_expression, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                     uint32_t bz, uint32_t dynamic_lds, void** params, void* stream) {
    GSE_REQUIRE(model && name_expression && params && gx >= 1 && gy >= 1 && gz >= 1 && bx >= 1 && by >= 1 && bz >= 1 &&
                    (uint64_t)bx * by * bz <= 1024, "gse_model_launch: bad argument");
    return launch_user_kernel(static_cast<UserModel*>(model), name_expression, dim3(gx, gy, gz), dim3(bx, by, bz), dynamic_lds,
                              (hipStream_t)stream, params);
}

int gse_model_destroy(void* model) {
    UserModel* m = static_cast<UserModel*>(model);
    if (!m) return GSE_OK;
    for (hipModule_t mod : m->modules) (void)hipModuleUnload(mod);
    delete m;
    return GSE_OK;
}

// Compile (or find) a kernel instance without launching it: lets a caller pay the compile time up front.
int gse_model_prepare(void* model, const char* name_expression) {
    GSE_REQUIRE(model && name_expression, "gse_model_prepare: null argument");
    return user_kernel(static_cast<UserModel*>(model), name_expression) ? GSE_OK : GSE_ERR_HIP;
}

// Compile only (no device needed, nothing is loaded or kept): bytes of code object, or a negative status.
int64_t gse_model_compile_check(void* model, const char* name_expression) {
    if (!model || !name_expression) return fail(GSE_ERR_ARG, "gse_model_compile_check: null argument");
    std::lock_guard<std::mutex> lock(g_mutex);
    std::vector<char> code;
    std::string lowered;
    if (int rc = compile_instance(static_cast<UserModel*>(model), name_expression, code, lowered)) return rc;
    return (int64_t)code.size();
}

int64_t gse_model_kernels(void* model) {
    return model ? (int64_t)static_cast<UserModel*>(model)->functions.size() : 0;
}

}  // extern "C"
