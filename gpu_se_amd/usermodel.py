"""User-supplied models for the particle filter.

The reference accepts ANY Python ``f(x, u, dt)`` / ``g(x, u)`` and lets Numba compile them for the device
(filter/particle.py:176-208; tests/ParticleFilter_test.py:6-20 is such a pair).  The counterpart here: a restricted form
of Python -- straight-line arithmetic -- is translated to HIP device functions and the particle-filter kernels are
compiled for them at run time (hipcc, once per model; the library is cached under gpu_se_amd/lib/user/ by the hash of
the generated source).  The kernels are the shipped ones: the fused step, resampling, moments; only the model inside
them changes.

Accepted form
    def f(x, u, dt):            # -> the INCREMENT of the state (the filter does x += f(x, u, dt) + noise, particle.py:265-277)
        x1, x2 = x              # unpacking, or x[0] / u[1] subscripts with constant indices
        u, = u
        dx1 = x1 - x2 * u       # assignments of expressions:  + - * / ** (constant power), unary -, numbers, names,
        ...                     #   a if cond else b, comparisons, min / max / abs / exp / log / sqrt / sin / cos / tanh / pow
        return dx1 * dt, dx2 * dt
    def g(x, u): ... return y1, y2
    1 <= Nx <= 5 states, 1 <= Nu <= 2 inputs, Ny = 2 outputs: the fast kernels (state-major rows of five, a two-dimensional
    measurement density).  Beyond that -- up to 8 states, 4 inputs, 1 .. 4 outputs (round 5) -- the dimension-generic kernels of
    csrc/gse_generic_kernels.h run the same algorithms (gpu_se_amd/filter/generic.py): correct and parallel, not tuned.
Anything else (loops, calls to other functions, attribute access other than numpy. / math.; more than 8 states) raises
``ModelError`` -- there is no CPU fallback.

Arithmetic: every expression is evaluated in fp64 on the fp32-stored state, without FMA contraction, and the new state is
rounded to fp32 once -- what the reference's device code does when the expression mixes array elements with Python
floats, and reproducible bit for bit by evaluating the same Python function on float64 copies of the fp32 state
(tests/test_gpu_usermodel.py).
"""
import ast
import ctypes
import hashlib
import inspect
import os
import shutil
import subprocess
import tempfile
import textwrap

from gpu_se_amd import _lib, build

MAX_NX, MAX_NU, NY = 5, 2, 2
_FUNCS = {"min": "fmin", "max": "fmax", "abs": "fabs", "exp": "exp", "log": "log", "sqrt": "sqrt", "sin": "sin",
          "cos": "cos", "tanh": "tanh", "pow": "pow", "fabs": "fabs", "minimum": "fmin", "maximum": "fmax",
          "absolute": "fabs", "power": "pow"}
_BINOPS = {ast.Add: "+", ast.Sub: "-", ast.Mult: "*", ast.Div: "/"}
_CMPOPS = {ast.Lt: "<", ast.LtE: "<=", ast.Gt: ">", ast.GtE: ">=", ast.Eq: "==", ast.NotEq: "!="}


class ModelError(NotImplementedError):
    pass


class _Translator:
    """One Python function -> a list of C++ statements and the expressions it returns."""

    def __init__(self, fn, param_roles, max_x=None, max_u=None):
        self.max_x = MAX_NX if max_x is None else max_x
        self.max_u = MAX_NU if max_u is None else max_u
        try:
            src = textwrap.dedent(inspect.getsource(fn))
        except (OSError, TypeError) as e:
            raise ModelError("the source of %r is not available (define it in a file): %s" % (fn, e))
        tree = ast.parse(src)
        defs = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
        if len(defs) != 1:
            raise ModelError("expected one plain function definition for %r" % (fn,))
        self.fn = defs[0]
        params = [a.arg for a in self.fn.args.args]
        if len(params) != len(param_roles) or self.fn.args.vararg or self.fn.args.kwarg:
            raise ModelError("%s must take exactly (%s)" % (self.fn.name, ", ".join(param_roles)))
        # name -> ("vector", role) | ("scalar", C expression)
        self.env = {}
        for name, role in zip(params, param_roles):
            self.env[name] = ("scalar", "dt_") if role == "dt" else ("vector", role)
        self.used = {"x": 0, "u": 0}
        self.lines = []
        self.counter = 0
        self.returns = None
        for stmt in self.fn.body:
            self.statement(stmt)
        if self.returns is None:
            raise ModelError("%s does not return anything" % self.fn.name)

    def fail(self, node, what):
        raise ModelError("%s, line %d: %s is outside the supported form (see gpu_se_amd.usermodel)"
                         % (self.fn.name, getattr(node, "lineno", 0), what))

    def element(self, role, index, node):
        limit = self.max_x if role == "x" else self.max_u
        if not 0 <= index < limit:
            self.fail(node, "%s[%d]" % (role, index))
        self.used[role] = max(self.used[role], index + 1)
        return "%s_%d" % (role, index)

    def fresh(self, hint):
        self.counter += 1
        return "v%d_%s" % (self.counter, "".join(c if c.isalnum() else "_" for c in hint))

    def statement(self, node):
        if self.returns is not None:
            self.fail(node, "code after return")
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Constant) and isinstance(node.value.value, str):
            return                                              # docstring
        if isinstance(node, ast.Return):
            v = node.value
            items = v.elts if isinstance(v, (ast.Tuple, ast.List)) else [v]
            self.returns = [self.expr(e) for e in items]
            return
        if isinstance(node, ast.AugAssign) and isinstance(node.target, ast.Name) and type(node.op) in _BINOPS:
            cur = self.expr(ast.Name(id=node.target.id, ctx=ast.Load()))
            self.bind(node.target.id, "(%s %s %s)" % (cur, _BINOPS[type(node.op)], self.expr(node.value)))
            return
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            target = node.targets[0]
            if isinstance(target, (ast.Tuple, ast.List)):
                names = target.elts
                if not all(isinstance(n, ast.Name) for n in names):
                    self.fail(node, "this unpacking")
                value = node.value
                if isinstance(value, ast.Name) and self.env.get(value.id, (None,))[0] == "vector":
                    role = self.env[value.id][1]
                    exprs = [self.element(role, i, node) for i in range(len(names))]
                elif isinstance(value, (ast.Tuple, ast.List)) and len(value.elts) == len(names):
                    exprs = [self.expr(e) for e in value.elts]
                else:
                    self.fail(node, "this unpacking")
                for n, e in zip(names, exprs):                  # (all right-hand sides were formed before any binding)
                    self.bind(n.id, e)
                return
            if isinstance(target, ast.Name):
                self.bind(target.id, self.expr(node.value))
                return
        self.fail(node, type(node).__name__)

    def bind(self, name, c_expr):
        var = self.fresh(name)
        self.lines.append("const double %s = %s;" % (var, c_expr))
        self.env[name] = ("scalar", var)

    def expr(self, node):
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return "(%s)" % repr(float(node.value))
        if isinstance(node, ast.Name):
            kind = self.env.get(node.id)
            if kind is None:
                self.fail(node, "the name %r" % node.id)
            if kind[0] == "vector":
                self.fail(node, "the whole vector %r in an expression" % node.id)
            return kind[1]
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and \
                self.env.get(node.value.id, (None,))[0] == "vector":
            idx = node.slice
            if isinstance(idx, ast.Constant) and isinstance(idx.value, int):
                return self.element(self.env[node.value.id][1], idx.value, node)
            self.fail(node, "a non-constant index")
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            return "(%s%s)" % ("-" if isinstance(node.op, ast.USub) else "+", self.expr(node.operand))
        if isinstance(node, ast.BinOp):
            if type(node.op) in _BINOPS:
                return "(%s %s %s)" % (self.expr(node.left), _BINOPS[type(node.op)], self.expr(node.right))
            if isinstance(node.op, ast.Pow):
                base = self.expr(node.left)
                p = node.right
                if isinstance(p, ast.Constant) and isinstance(p.value, int) and 1 <= p.value <= 8:
                    # the repeated products CPython's float.__pow__ does NOT do: keep pow() semantics exact instead
                    return "pow(%s, %s)" % (base, repr(float(p.value)))
                return "pow(%s, %s)" % (base, self.expr(p))
        if isinstance(node, ast.IfExp):
            return "((%s) ? %s : %s)" % (self.condition(node.test), self.expr(node.body), self.expr(node.orelse))
        if isinstance(node, ast.Call) and not node.keywords:
            f = node.func
            name = f.id if isinstance(f, ast.Name) else (
                f.attr if isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name)
                and f.value.id in ("numpy", "np", "math") else None)
            if name in _FUNCS:
                args = [self.expr(a) for a in node.args]
                c = _FUNCS[name]
                if c in ("fmin", "fmax") and len(args) > 2:
                    out = args[0]
                    for a in args[1:]:
                        out = "%s(%s, %s)" % (c, out, a)
                    return out
                want = 2 if c in ("fmin", "fmax", "pow") else 1
                if len(args) != want:
                    self.fail(node, "%s with %d arguments" % (name, len(args)))
                return "%s(%s)" % (c, ", ".join(args))
            self.fail(node, "a call to %s" % ast.unparse(f))
        self.fail(node, ast.unparse(node) if hasattr(ast, "unparse") else type(node).__name__)

    def condition(self, node):
        if isinstance(node, ast.Compare) and len(node.ops) == 1 and type(node.ops[0]) in _CMPOPS:
            return "%s %s %s" % (self.expr(node.left), _CMPOPS[type(node.ops[0])], self.expr(node.comparators[0]))
        if isinstance(node, ast.BoolOp):
            op = " && " if isinstance(node.op, ast.And) else " || "
            return op.join("(%s)" % self.condition(v) for v in node.values)
        self.fail(node, "this condition")


def translate(f, g):
    """(nx, nu, header text) for a pair of Python model functions."""
    tf, tg = _Translator(f, ("x", "u", "dt")), _Translator(g, ("x", "u"))
    nx = len(tf.returns)
    if not 1 <= nx <= MAX_NX or max(tf.used["x"], tg.used["x"]) > nx:
        raise ModelError("f must return one increment per state (1..%d states); it returns %d and the functions read "
                         "%d states" % (MAX_NX, nx, max(tf.used["x"], tg.used["x"])))
    if len(tg.returns) != NY:
        raise ModelError("g must return exactly %d outputs (the measurement density of the kernels is %d-dimensional); "
                         "it returns %d" % (NY, NY, len(tg.returns)))
    nu = max(tf.used["u"], tg.used["u"], 1)

    def prologue(cast):
        return ["const double x_%d = %s x_[%d];" % (k, cast, k) for k in range(MAX_NX)] + \
               ["const double u_0 = u0_, u_1 = u1_;", "(void)x_0; (void)x_1; (void)x_2; (void)x_3; (void)x_4; (void)u_0; (void)u_1;"]

    body_f = prologue("(double)") + tf.lines + \
        ["dx_[%d] = %s;" % (k, tf.returns[k] if k < nx else "0.0") for k in range(MAX_NX)]
    body_g = prologue("(double)") + tg.lines + ["y0_ = (float)(%s);" % tg.returns[0], "y1_ = (float)(%s);" % tg.returns[1]]
    text = "\n".join([
        "// generated by gpu_se_amd/usermodel.py from %s / %s -- do not edit" % (f.__name__, g.__name__),
        "#pragma once",
        "#define GSE_USER_NX %d" % nx,
        "#define GSE_USER_NU %d" % nu,
        "template <class T>",
        "__device__ __forceinline__ void user_increment(const T (&x_)[5], double u0_, double u1_, double dt_, double (&dx_)[5]) {",
        "#pragma clang fp contract(off)",
        "    " + "\n    ".join(body_f),
        "}",
        "__device__ __forceinline__ void user_output(const float (&x_)[5], double u0_, double u1_, float& y0_, float& y1_) {",
        "#pragma clang fp contract(off)",
        "    const double dt_ = 0.0; (void)dt_;",
        "    " + "\n    ".join(body_g),
        "}", ""])
    return nx, nu, text


G_MAX_NX, G_MAX_NU, G_MAX_NY = 8, 4, 4


def model_dimensions(f, g):
    """(nx, nu, ny) of a pair of model functions (translated with the generic bounds)."""
    tf, tg = _Translator(f, ("x", "u", "dt"), G_MAX_NX, G_MAX_NU), _Translator(g, ("x", "u"), G_MAX_NX, G_MAX_NU)
    return len(tf.returns), max(tf.used["u"], tg.used["u"], 1), len(tg.returns)


def needs_generic_kernels(f, g):
    """True when (f, g) is outside the fast kernels' bounds (5 states, 2 inputs, exactly 2 outputs) but inside the generic
    ones'; outside both: ModelError."""
    nx, nu, ny = model_dimensions(f, g)
    if nx <= MAX_NX and nu <= MAX_NU and ny == NY:
        return False
    if not (1 <= nx <= G_MAX_NX and 1 <= nu <= G_MAX_NU and 1 <= ny <= G_MAX_NY):
        raise ModelError("a model of %d states, %d inputs, %d outputs is beyond the generic kernels' bounds (%d, %d, %d)"
                         % (nx, nu, ny, G_MAX_NX, G_MAX_NU, G_MAX_NY))
    return True


def translate_generic(f, g):
    """(nx, nu, ny, header text) for the dimension-generic kernels (csrc/gse_generic_kernels.h)."""
    tf, tg = _Translator(f, ("x", "u", "dt"), G_MAX_NX, G_MAX_NU), _Translator(g, ("x", "u"), G_MAX_NX, G_MAX_NU)
    nx, ny = len(tf.returns), len(tg.returns)
    nu = max(tf.used["u"], tg.used["u"], 1)
    if not 1 <= nx <= G_MAX_NX or max(tf.used["x"], tg.used["x"]) > nx:
        raise ModelError("f must return one increment per state (1..%d states); it returns %d and the functions read %d states"
                         % (G_MAX_NX, nx, max(tf.used["x"], tg.used["x"])))
    if not 1 <= ny <= G_MAX_NY:
        raise ModelError("g must return 1..%d outputs; it returns %d" % (G_MAX_NY, ny))
    names = ["x_%d" % k for k in range(nx)] + ["u_%d" % k for k in range(nu)]
    prologue = ["const double x_%d = x_[%d];" % (k, k) for k in range(nx)] + \
               ["const double u_%d = u_[%d];" % (k, k) for k in range(nu)] + [" ".join("(void)%s;" % n for n in names)]
    body_f = prologue + tf.lines + ["dx_[%d] = %s;" % (k, tf.returns[k]) for k in range(nx)]
    body_g = prologue + tg.lines + ["y_[%d] = (float)(%s);" % (k, tg.returns[k]) for k in range(ny)]
    text = "\n".join([
        "// generated by gpu_se_amd/usermodel.py from %s / %s (generic kernels) -- do not edit" % (f.__name__, g.__name__),
        "#pragma once",
        "#define GSE_G_NX %d" % nx, "#define GSE_G_NU %d" % nu, "#define GSE_G_NY %d" % ny,
        "__device__ __forceinline__ void generic_increment(const double (&x_)[%d], const double (&u_)[%d], double dt_, "
        "double (&dx_)[%d]) {" % (nx, nu, nx),
        "#pragma clang fp contract(off)",
        "    " + "\n    ".join(body_f),
        "}",
        "__device__ __forceinline__ void generic_output(const double (&x_)[%d], const double (&u_)[%d], float (&y_)[%d]) {"
        % (nx, nu, ny),
        "#pragma clang fp contract(off)",
        "    const double dt_ = 0.0; (void)dt_;",
        "    " + "\n    ".join(body_g),
        "}", ""])
    return nx, nu, ny, text


class GenericModel:
    """A user model beyond the fast kernels' bounds: its kernels (csrc/gse_generic_kernels.h) are compiled in the process by
    hiprtc with the model's dimensions as constants, one kernel at a time on first launch, and launched by name."""

    def __init__(self, f, g):
        import torch
        self.nx, self.nu, self.ny, text = translate_generic(f, g)
        self.text = text
        h = hashlib.sha256(text.encode())
        with open(os.path.join(build.CSRC, "gse_generic_kernels.h"), "rb") as fh:
            h.update(fh.read())
        self.tag = int(h.hexdigest()[:15], 16) | 1
        base = _lib.load()
        handle = ctypes.c_void_p()
        rtc = os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so")
        if not os.path.exists(rtc):
            rtc = "/opt/rocm/lib/libhiprtc.so"
        _lib.check(base.gse_generic_model_create(text.encode(), self.nx, self.nu, self.ny, self.tag, build.CSRC.encode(),
                                                 os.path.join(build.ROOT, "include").encode(), rtc.encode(), ctypes.byref(handle)))
        self._base, self._handle = base, handle

    def launch(self, name, n_threads, block, args, stream):
        """Launch kernel ``name`` (a C++ name inside gse::) over n_threads threads in workgroups of ``block``; args: ctypes
        objects, one per kernel argument (kept alive by the caller until the call returns -- the launch copies them)."""
        params = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p).value for a in args])
        grid = max(1, -(-int(n_threads) // block))
        _lib.check(self._base.gse_model_launch(self._handle, ("gse::" + name).encode(), grid, 1, 1, block, 1, 1, 0, params, stream))

    def compile_check(self, name):
        """Compile only (works without a device): size of the code object in bytes."""
        n = self._base.gse_model_compile_check(self._handle, ("gse::" + name).encode())
        if n < 0:
            _lib.check(int(n))
        return int(n)

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle is not None and getattr(self, "_base", None) is not None:
            try:
                self._base.gse_model_destroy(handle)
            except Exception:
                pass


_generic_cache = {}


def compile_generic_model(f, g):
    key = (f, g)
    if key not in _generic_cache:
        _generic_cache[key] = GenericModel(f, g)
    return _generic_cache[key]


USER_DIR = os.path.join(build.LIBDIR, "user")
USER_SOURCES = ["gse_host.hip", "gse_pf.hip", "gse_gsukf.hip", "gse_rows.hip", "gse_rtc.hip"]     # everything the filters call


class CompiledModel:
    """A user model's library: the C ABI of include/gse.h restricted to the filters' entry points."""

    def __init__(self, nx, nu, tag, path):
        self.nx, self.nu, self.tag, self.path = nx, nu, tag, path
        import torch  # noqa: F401  (its HIP runtime must be the one in the process, see _lib.load)
        self.lib = ctypes.CDLL(path)
        for name, (res, args) in _lib.PROTOTYPES.items():
            fn = getattr(self.lib, name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
        if int(self.lib.gse_model_tag()) != tag:
            raise ImportError("%s was not built for this model" % path)

    def check(self, rc):
        if rc != 0:
            msg = self.lib.gse_last_error()
            raise _lib.GseError("libgse_user: error %d: %s" % (rc, msg.decode() if msg else "?"))


class _ModelEntryPoints:
    """The C ABI as a filter with a run-time compiled model sees it: the step, the GS-UKF calls and the row helpers go to their
    ``*_model`` forms with the model handle in front, everything else (workspace, resample_begin, moments, draws -- none of
    which contains the model) is the shipped library's."""
    _ROUTED = ("gse_pf_step", "gse_model_f", "gse_model_g_u", "gse_gsukf_sigma_points", "gse_gsukf_predict", "gse_gsukf_update",
               "gse_gsukf_predict_update", "gse_gsukf_cycle", "gse_gsukf_resample", "gse_gsukf_moments", "gse_gsukf_gather_rows",
               "gse_gsukf_predict_update_from")

    def __init__(self, lib, handle):
        self._lib, self._handle = lib, handle

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self._ROUTED:
            routed = getattr(self._lib, name + "_model")
            handle = self._handle
            return lambda *args: routed(handle, *args)
        return fn


class RuntimeModel:
    """A user model whose kernels are compiled IN THE PROCESS by hiprtc, one template instance at a time, the first time a
    launch needs it (about a second each; gse_rtc.hip): no compiler binary, no files, nothing to pre-build."""

    def __init__(self, nx, nu, tag, text):
        import torch
        self.nx, self.nu, self.tag, self.path = nx, nu, tag, None
        base = _lib.load()
        handle = ctypes.c_void_p()
        # the hiprtc that belongs to the HIP runtime in the process (torch's), else ROCm's
        rtc = os.path.join(os.path.dirname(torch.__file__), "lib", "libhiprtc.so")
        if not os.path.exists(rtc):
            rtc = "/opt/rocm/lib/libhiprtc.so"
        _lib.check(base.gse_model_create(text.encode(), nx, nu, tag, build.CSRC.encode(),
                                         os.path.join(build.ROOT, "include").encode(), rtc.encode(), ctypes.byref(handle)))
        self._base, self._handle = base, handle
        self.lib = _ModelEntryPoints(base, handle)

    def check(self, rc):
        _lib.check(rc)

    def __del__(self):
        # the loaded code objects belong to the handle: unload them with it (compile_model caches models for the life of the
        # process, so this runs for models created directly and at interpreter exit)
        handle, self._handle = getattr(self, "_handle", None), None
        if handle is not None and getattr(self, "_base", None) is not None:
            try:
                self._base.gse_model_destroy(handle)
            except Exception:
                pass

    def prepare(self, name_expression):
        """Compile a kernel instance now, e.g. "gse::pf_step_kernel<true, true, false, true, 2, 3>" (needs the device)."""
        _lib.check(self._base.gse_model_prepare(self._handle, name_expression.encode()))

    def compile_check(self, name_expression):
        """Compile only (works without a device): size of the code object in bytes."""
        n = self._base.gse_model_compile_check(self._handle, name_expression.encode())
        if n < 0:
            _lib.check(int(n))
        return int(n)

    @property
    def kernels(self):
        return int(self._base.gse_model_kernels(self._handle))


_cache = {}


def compile_model(f, g, verbose=False, backend=None):
    """Translate (f, g) and make their kernels available.  backend "rtc" (default; GSE_USERMODEL_BACKEND overrides): run-time
    compilation in the process, on demand (RuntimeModel).  "hipcc": the whole particle-filter library rebuilt for the model
    by the hipcc binary into gpu_se_amd/lib/user/<hash>/ (~13 s once, then cached on disk; CompiledModel)."""
    backend = backend or os.environ.get("GSE_USERMODEL_BACKEND", "rtc")
    if backend not in ("rtc", "hipcc"):
        raise ValueError("unknown user-model backend %r" % backend)
    key = (f, g, backend)
    if key in _cache:
        return _cache[key]
    nx, nu, text = translate(f, g)
    # the cache key covers everything the library is made of: the generated model, every source it compiles, every header
    # they include (workspace layout, struct ABI) and the ABI version
    h = hashlib.sha256(text.encode())
    for name in sorted(set(USER_SOURCES) | {n for n in os.listdir(build.CSRC) if n.endswith(".h")}):
        with open(os.path.join(build.CSRC, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    with open(os.path.join(build.ROOT, "include", "gse.h"), "rb") as fh:
        h.update(b"gse.h\0" + fh.read())
    digest = h.hexdigest()
    if backend == "rtc":
        model = RuntimeModel(nx, nu, int(digest[:15], 16) | 1, text)
        _cache[key] = model
        return model
    tag = int(digest[:15], 16) | 1                                       # non-zero, fits a signed 64-bit literal
    folder = os.path.join(USER_DIR, digest[:16])
    path = os.path.join(folder, "libgse_user.so")
    if not os.path.exists(path):
        os.makedirs(folder, exist_ok=True)
        final_folder = folder
        folder = tempfile.mkdtemp(prefix="build%d_" % os.getpid(), dir=final_folder)   # ranks building the same model at once
        header = os.path.join(folder, "gse_user_model.h")                              # never share a half-written file
        with open(header, "w") as fh:
            fh.write(text)
        flags = ["-DGSE_USER_MODEL=\"%s\"" % header, "-DGSE_USER_MODEL_TAG=%dull" % tag]
        objs, procs = [], []
        for src in USER_SOURCES:
            obj = os.path.join(folder, src.replace(".hip", ".o"))
            cmd = [build._hipcc(), "--offload-arch=" + build.ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                   "-Wno-unused-variable", "-I" + os.path.join(build.ROOT, "include"), "-I" + build.CSRC] + flags + \
                  ["-c", os.path.join(build.CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
            objs.append(obj)
        for src, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise ModelError("hipcc failed on %s for the user model:\n%s\n--- generated source ---\n%s" % (src, out, text))
        tmp = os.path.join(folder, "libgse_user.so")
        subprocess.run([build._hipcc(), "--offload-arch=" + build.ARCH, "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"], check=True)
        os.replace(tmp, path)                                                          # atomic: readers see all or nothing
        shutil.rmtree(folder, ignore_errors=True)
    model = CompiledModel(nx, nu, tag, path)
    _cache[key] = model
    return model
