"""ctypes binding of libmasp_hip.so (C ABI: include/masp_hip.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

ERRORS = {1: "invalid argument", 2: "Parameters bytes are malformed", 3: "Parameters do not match the circuit shape",
          4: "no usable HIP device (there is no CPU fallback)", 5: "HIP runtime error", 6: "UnexpectedIdentity",
          7: "circuit slot is empty", 8: "scalar is not a canonical field element"}

SPEND, OUTPUT, CONVERT = 0, 1, 2


class MaspHipError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("masp_hip error %d: %s%s" % (code, ERRORS.get(code, "?"), (" — " + detail) if detail else ""))


OPTION_FIELDS = ("slots", "batch_cap", "ntt_sub_batch", "window_bits_h", "window_bits_la", "window_bits_b", "window_bits_b2_lone",
                 "witness_nontrivial_percent", "bucket_tree_levels", "bucket_tree_sub_batch", "bucket_tree_levels_g2", "bucket_tree_scratch_mb", "lone_proof_graph", "window_bits_h_lone",
                 "window_bits_b2")


class OptionsStruct(C.Structure):
    """masp_hip_options (include/masp_hip.h): every field 0 = the default."""
    _fields_ = ([("struct_size", C.c_uint32)] + [(f, C.c_int32) for f in OPTION_FIELDS[:OPTION_FIELDS.index("lone_proof_graph")]] +
                [("bucket_tree_fallback_proofs", C.c_int32), ("lone_proof_graph", C.c_int32), ("hw_queues", C.c_int32), ("window_bits_h_lone", C.c_int32),
                 ("digit_recoding", C.c_int32),     # reserved since round 6 (was: NAF digits over per-bit tables), must be 0
                 ("window_bits_b2", C.c_int32)])


assert C.sizeof(OptionsStruct) == 76 and OptionsStruct.window_bits_b2.offset == 72        # include/masp_hip.h asserts the same


class JobStruct(C.Structure):
    _fields_ = [("circuit", C.c_uint32), ("inputs", C.c_void_p), ("aux", C.c_void_p), ("a", C.c_void_p),
                ("b", C.c_void_p), ("c", C.c_void_p), ("r", C.c_uint8 * 32), ("s", C.c_uint8 * 32), ("aux_form", C.c_uint32),
                ("reserved", C.c_uint32)]


assert C.sizeof(JobStruct) == 120 and JobStruct.r.offset == 48 and JobStruct.aux_form.offset == 112   # include/masp_hip.h asserts the same
AUX_CANONICAL, AUX_MONTGOMERY = 0, 1     # masp_hip_job::aux_form


def library_path():
    # MASP_HIP_LIBRARY: an alternative build of the same library (A/B measurements of compile-time parameters)
    return os.environ.get("MASP_HIP_LIBRARY") or os.path.join(_HERE, "libmasp_hip.so")


def load_library():
    """Loads the HIP extension; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError("libmasp_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C masp_amd/csrc` (hipcc, --offload-arch=gfx950)")
    # batches in flight on different HIP streams only overlap if the runtime gives them their own hardware queues
    # (ROCm's default is 4); read once, when the HIP runtime initialises.  16: the slots' main streams and the context's own need a queue each; with 8
    # queues about one bench process in four landed in a mode 3 % slower (streams of two slots sharing a queue), with 16 none of
    # twelve did (profiles/r04e_hw_queues_and_the_two_modes.txt)
    # (the library's constructor does the same for a process that links it directly — masp_hip_runtime_prepare, include/masp_hip.h —;
    # here it is said before the load because an A/B run may load an older build.  What the runtime really uses is MEASURED per context:
    # Context.options["hw_queues"])
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    L = C.CDLL(path)
    if hasattr(L, "masp_hip_runtime_prepare"):
        L.masp_hip_runtime_prepare.argtypes = [C.c_int, C.c_int]
        # more than 20 hardware queues make a process SLOWER once its runtime has created them (the cap is what counts: the pool only
        # grows) — masp_hip_runtime_prepare(0, 0) changes nothing and returns what the environment says
        q = L.masp_hip_runtime_prepare(0, 0)
        if q > 20:
            import warnings
            warnings.warn("masp_amd: GPU_MAX_HW_QUEUES=%d: once this process holds more than 20 hardware queues every kernel dispatch gets "
                          "slower (24: -7 %%, 32: -21 %% proofs/s; profiles/r06_second_context_root_cause.txt) — use 16" % q)
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    L.masp_hip_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.masp_hip_ctx_create_multi.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.masp_hip_ctx_create_ex.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(OptionsStruct), C.POINTER(vp)]
    L.masp_hip_ctx_get_options.argtypes = [vp, C.POINTER(OptionsStruct)]
    L.masp_hip_device_count.argtypes = []
    L.masp_hip_options_default.argtypes = [C.POINTER(OptionsStruct)]
    L.masp_hip_options_default.restype = None
    L.masp_hip_ctx_device_count.argtypes = [vp]
    L.masp_hip_ctx_device_proofs.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    if hasattr(L, "masp_hip_ctx_stream_concurrency"):
        L.masp_hip_ctx_stream_concurrency.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if hasattr(L, "masp_hip_ctx_device_status"):            # (round 6; an older build passed as MASP_HIP_LIBRARY lacks them)
        L.masp_hip_ctx_device_status.argtypes = [vp, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_uint64)]
        L.masp_hip_ctx_inject_fault.argtypes = [vp, C.c_int, C.c_uint32]
    # (introspection added in round 4: an older build passed as MASP_HIP_LIBRARY for an A/B run lacks them; calling them then raises)
    if hasattr(L, "masp_hip_ctx_lone_graph_launches"):
        L.masp_hip_ctx_lone_graph_launches.argtypes = [vp, C.POINTER(C.c_uint64)]
    if hasattr(L, "masp_hip_circuit_flags"):
        L.masp_hip_circuit_flags.argtypes = [vp, u32, C.POINTER(C.c_uint32)]
    L.masp_hip_ctx_destroy.argtypes = [vp]
    L.masp_hip_ctx_destroy.restype = None
    L.masp_hip_strerror.restype = C.c_char_p
    L.masp_hip_last_error.argtypes = [vp]
    L.masp_hip_last_error.restype = C.c_char_p
    L.masp_hip_circuit_load.argtypes = [vp, u32, vp, sz, vp]
    L.masp_hip_prove.argtypes = [vp, u32, vp, vp, vp, vp, vp, C.c_char_p, C.c_char_p, vp]
    L.masp_hip_prove_batch.argtypes = [vp, sz, vp, vp]
    L.masp_hip_generate_parameters.argtypes = [vp, vp, C.c_char_p, vp, sz, C.POINTER(sz)]
    L.masp_hip_parameters_max_size.argtypes = [vp]
    L.masp_hip_parameters_max_size.restype = sz
    L.masp_hip_msm_g1.argtypes = [vp, vp, vp, sz, vp]
    L.masp_hip_msm_g2.argtypes = [vp, vp, vp, sz, vp]
    L.masp_hip_msm_g1_multi.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp]
    L.masp_hip_msm_g2_multi.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp]
    L.masp_hip_quotient_h.argtypes = [vp, vp, vp, vp, sz, u32, vp]
    L.masp_hip_ntt.argtypes = [vp, vp, u32, C.c_int]
    L.masp_hip_vk_prepare.argtypes = [vp, vp, sz, C.POINTER(vp)]
    L.masp_hip_vk_free.argtypes = [vp]
    L.masp_hip_vk_free.restype = None
    L.masp_hip_verify_batch.argtypes = [vp, vp, sz, vp, vp, u32, vp, C.POINTER(C.c_int)]
    L.masp_hip_batch_upload.argtypes = [vp, sz, vp]
    L.masp_hip_batch_prove_resident.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_float)]
    L.masp_hip_batch_prove_resident_steps.argtypes = [vp, C.c_int, sz, vp, vp, C.POINTER(C.c_float)]
    L.masp_hip_batch_free.argtypes = [vp, C.c_int]
    L.masp_hip_bench_msm.argtypes = [vp, C.c_int, sz, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(u32)]
    L.masp_hip_profile_enable.argtypes = [vp, C.c_int]
    L.masp_hip_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.masp_hip_profile_read_split.argtypes = [vp, C.POINTER(C.c_double)]
    if hasattr(L, "masp_hip_profile_read_lone"):
        L.masp_hip_profile_read_lone.argtypes = [vp, C.POINTER(C.c_double)]
    L.masp_hip_sync.argtypes = [vp]
    L.masp_hip_host_alloc.argtypes = [vp, sz]
    L.masp_hip_host_alloc.restype = vp
    L.masp_hip_host_free.argtypes = [vp, vp]
    L.masp_hip_host_free.restype = None
    _lib = L
    return L


def device_pci_bus_id(device=0):
    """masp_hip_device_pci_bus_id -> "0000:c1:00.0" (lower case), or None"""
    L = load_library()
    if not hasattr(L, "masp_hip_device_pci_bus_id"):
        return None
    buf = C.create_string_buffer(64)
    L.masp_hip_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    return buf.value.decode().lower() if L.masp_hip_device_pci_bus_id(int(device), buf, 64) == 0 else None


def device_count():
    """HIP devices visible to this process (masp_hip_device_count)."""
    return int(load_library().masp_hip_device_count())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a, shape_last=None):
    if a is None:
        return None
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(a, dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape_last is not None:
        a = a.reshape(-1, shape_last)
    return a


def _scalar32(x):
    if isinstance(x, int):
        return x.to_bytes(32, "little")
    b = bytes(x)
    assert len(b) == 32
    return b


class Context:
    """One GPU (`device` an int) or several GPUs of one node behind one prover (`device` a list of device indices:
    masp_hip_ctx_create_multi — prove_batch then deals its jobs to the devices).  Thread-safe: prove / prove_batch are
    re-entrant, everything else serialises inside the library."""

    def __init__(self, device=0, **options):
        """options: the fields of masp_hip_options (slots, batch_cap, ntt_sub_batch, window_bits_*, ...); unset = default.
        The library reads no environment variables: tuning comes through here."""
        self._L = load_library()
        h = C.c_void_p()
        unknown = set(options) - set(OPTION_FIELDS)
        if unknown:
            raise TypeError("unknown context option(s): %s" % ", ".join(sorted(unknown)))
        opt = OptionsStruct()
        self._L.masp_hip_options_default(C.byref(opt))
        for k, v in options.items():
            if v is not None:
                setattr(opt, k, int(v))
        if isinstance(device, (list, tuple)):
            arr = (C.c_int * len(device))(*[int(d) for d in device])
            if options or len(device) > 1:
                rc = self._L.masp_hip_ctx_create_ex(arr, len(device), C.byref(opt), C.byref(h))
            else:
                rc = self._L.masp_hip_ctx_create_multi(arr, len(device), C.byref(h))
        else:
            arr = (C.c_int * 1)(int(device))
            rc = self._L.masp_hip_ctx_create_ex(arr, 1, C.byref(opt), C.byref(h))
        if rc:
            raise MaspHipError(rc)
        got = OptionsStruct()
        got.struct_size = C.sizeof(OptionsStruct)           # the library writes no more than the caller's struct holds
        self._L.masp_hip_ctx_get_options(h, C.byref(got))
        self.options = {f: int(getattr(got, f)) for f in OPTION_FIELDS}      # defaults resolved
        # hardware queues the runtime spreads this process's streams over, measured at creation (masp_hip_options::hw_queues): fewer than
        # the slots have streams (five each) means independent kernels of a batch wait for each other — the HIP runtime was initialised
        # (by whatever this process loaded first) before GPU_MAX_HW_QUEUES said 16
        self.options["hw_queues"] = int(got.hw_queues)
        want = min(5 * self.options["slots"], 15)
        if 0 < self.options["hw_queues"] < want:
            import warnings
            warnings.warn("masp_amd: the HIP runtime gives this process %d hardware queue(s) for %d slots x 5 streams: set GPU_MAX_HW_QUEUES=16 "
                          "before the process's first HIP call (masp_hip_runtime_prepare)" % (self.options["hw_queues"], self.options["slots"]))

        self._h = h
        self._keep = []
        self._pinned = {}

    def close(self):
        if getattr(self, "_h", None):
            for p in list(getattr(self, "_pinned", {}).values()):
                self._L.masp_hip_host_free(self._h, p)
            self._pinned = {}
            self._L.masp_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise MaspHipError(rc, self._L.masp_hip_last_error(self._h).decode(errors="replace"))

    # ---- page-locked host buffers (assignments written here reach the device without a staging copy) ----
    def host_alloc(self, rows, cols=32):
        """-> np.uint8 array [rows, cols] over memory from masp_hip_host_alloc; release with host_free(array)."""
        p = self._L.masp_hip_host_alloc(self._h, rows * cols)
        if not p:
            raise MemoryError("masp_hip_host_alloc(%d bytes) failed" % (rows * cols))
        buf = (C.c_uint8 * (rows * cols)).from_address(p)
        a = np.frombuffer(buf, dtype=np.uint8).reshape(rows, cols)
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = self._pinned.pop(a.ctypes.data, None)
        if p and getattr(self, "_h", None):
            self._L.masp_hip_host_free(self._h, p)

    # ---- circuits / proofs ----
    def load_circuit(self, slot, params, cs):
        params = _u8(params)
        self._check(self._L.masp_hip_circuit_load(self._h, slot, _p(params), params.size, cs.ref))

    def _job(self, slot, inputs, aux, r, s, abc=None, aux_form=AUX_CANONICAL):
        j = JobStruct()
        j.circuit = slot
        j.aux_form = int(aux_form)
        inputs, aux = _u8(inputs, 32), _u8(aux, 32)
        keep = [inputs, aux]
        j.inputs, j.aux = inputs.ctypes.data, aux.ctypes.data
        if abc is not None:
            a, b, c = (_u8(x, 32) for x in abc)
            keep += [a, b, c]
            j.a, j.b, j.c = a.ctypes.data, b.ctypes.data, c.ctypes.data
        j.r[:] = _scalar32(r)
        j.s[:] = _scalar32(s)
        return j, keep

    def prove(self, slot, inputs, aux, r, s, abc=None):
        """-> 192-byte proof (A | B | C compressed)."""
        return self.prove_batch([(slot, inputs, aux, r, s, abc)])[0]

    def marshal_jobs(self, jobs):
        """jobs: iterable of (slot, inputs, aux, r, s[, (a,b,c)[, aux_form]]) -> (masp_hip_job array, n, keep-alive list): the argument
        of masp_hip_prove_batch, built once when the same list is proved repeatedly or timed."""
        jobs = list(jobs)
        arr = (JobStruct * len(jobs))()
        keep = []
        for i, job in enumerate(jobs):
            slot, inputs, aux, r, s = job[:5]
            abc = job[5] if len(job) > 5 else None
            arr[i], k = self._job(slot, inputs, aux, r, s, abc, job[6] if len(job) > 6 else AUX_CANONICAL)
            keep.append(k)
        return arr, len(jobs), keep

    def prove_marshalled(self, arr, n, out=None):
        """One masp_hip_prove_batch call -> u8[n,192]"""
        if out is None:
            out = np.zeros((n, 192), dtype=np.uint8)
        self._check(self._L.masp_hip_prove_batch(self._h, n, arr, _p(out)))
        return out

    def prove_batch(self, jobs):
        """jobs: iterable of (slot, inputs, aux, r, s[, (a,b,c)]) -> list of 192-byte proofs, job order."""
        arr, n, keep = self.marshal_jobs(jobs)
        out = self.prove_marshalled(arr, n)
        return [out[i].tobytes() for i in range(n)]

    def generate_parameters(self, cs, toxic):
        """toxic: (tau, alpha, beta, gamma, delta) ints -> Parameters bytes (bellman wire format), np.uint8."""
        t = b"".join(_scalar32(x) for x in toxic)
        cap = self._L.masp_hip_parameters_max_size(cs.ref)
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._check(self._L.masp_hip_generate_parameters(self._h, cs.ref, t, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    # ---- Groth16 batch verification on the GPU ----
    def prepare_verifying_key(self, params):
        """-> GpuVerifyingKey (= PreparedVerifyingKey, masp_proofs/src/lib.rs:391-393, for masp_hip_verify_batch)"""
        return GpuVerifyingKey(self, params)

    # ---- building blocks ----
    def msm_g1(self, bases, scalars):
        bases, scalars = _u8(bases, 96), _u8(scalars, 32)
        out = np.zeros(96, dtype=np.uint8)
        self._check(self._L.masp_hip_msm_g1(self._h, _p(bases), _p(scalars), scalars.shape[0], _p(out)))
        return out.tobytes()

    def msm_g2(self, bases, scalars):
        bases, scalars = _u8(bases, 192), _u8(scalars, 32)
        out = np.zeros(192, dtype=np.uint8)
        self._check(self._L.masp_hip_msm_g2(self._h, _p(bases), _p(scalars), scalars.shape[0], _p(out)))
        return out.tobytes()

    def msm_g1_multi(self, bases, scalars, window_bits=0):
        """bases u8[n,96]; scalars u8[np,n,32] -> list of np 96-byte results (one batched launch sequence)."""
        bases = _u8(bases, 96)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        npf, n = scalars.shape[0], scalars.shape[1]
        assert scalars.shape == (npf, n, 32) and bases.shape[0] == n
        out = np.zeros((npf, 96), dtype=np.uint8)
        self._check(self._L.masp_hip_msm_g1_multi(self._h, _p(bases), n, _p(scalars), npf, int(window_bits), _p(out)))
        return [out[i].tobytes() for i in range(npf)]

    def msm_g2_multi(self, bases, scalars, window_bits=0):
        """bases u8[n,192]; scalars u8[np,n,32] -> list of np 192-byte results (one batched launch sequence)."""
        bases = _u8(bases, 192)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
        npf, n = scalars.shape[0], scalars.shape[1]
        assert scalars.shape == (npf, n, 32) and bases.shape[0] == n
        out = np.zeros((npf, 192), dtype=np.uint8)
        self._check(self._L.masp_hip_msm_g2_multi(self._h, _p(bases), n, _p(scalars), npf, int(window_bits), _p(out)))
        return [out[i].tobytes() for i in range(npf)]

    def current_options(self):
        """masp_hip_ctx_get_options now: the options with what lack of tree scratch has changed since creation — the
        `bucket_tree_sub_batch` in use and `bucket_tree_fallback_proofs` (proofs that went through the XYZZ accumulation)."""
        got = OptionsStruct()
        got.struct_size = C.sizeof(OptionsStruct)
        self._check(self._L.masp_hip_ctx_get_options(self._h, C.byref(got)))
        d = {f: int(getattr(got, f)) for f in OPTION_FIELDS}
        d["bucket_tree_fallback_proofs"] = int(got.bucket_tree_fallback_proofs)
        d["hw_queues"] = int(got.hw_queues)
        return d

    @property
    def device_count(self):
        return self._L.masp_hip_ctx_device_count(self._h)

    def circuit_uses_endomorphism(self, slot):
        """masp_hip_circuit_flags bit 0: every CRS point behind A and B1 is in the prime-order subgroup"""
        f = C.c_uint32(0)
        self._check(self._L.masp_hip_circuit_flags(self._h, int(slot), C.byref(f)))
        return bool(f.value & 1)

    def lone_graph_launches(self):
        """small batches replayed from a captured launch graph so far (masp_hip_ctx_lone_graph_launches)"""
        v = C.c_uint64(0)
        self._check(self._L.masp_hip_ctx_lone_graph_launches(self._h, C.byref(v)))
        return int(v.value)

    def device_proofs(self):
        """proofs written so far by each device context of this prover (masp_hip_ctx_device_proofs)"""
        n = self.device_count
        counts = (C.c_uint64 * n)()
        self._check(self._L.masp_hip_ctx_device_proofs(self._h, counts, n))
        return list(counts)

    def stream_concurrency(self, mains_only=False):
        """masp_hip_ctx_stream_concurrency -> (streams of this context — or only the ones that carry batches —, how many of them ran a kernel at
        the same time)"""
        n, c = C.c_int(0), C.c_int(0)
        self._check(self._L.masp_hip_ctx_stream_concurrency(self._h, 1 if mains_only else 0, C.byref(n), C.byref(c)))
        return int(n.value), int(c.value)

    def device_status(self):
        """masp_hip_ctx_device_status -> ([0 or the error code that took device context d out], proofs put back on the queue so far)"""
        n = self.device_count
        st, rq = (C.c_int32 * n)(), C.c_uint64(0)
        self._check(self._L.masp_hip_ctx_device_status(self._h, st, n, C.byref(rq)))
        return list(st), int(rq.value)

    def inject_fault(self, device, nth):
        """TEST HOOK masp_hip_ctx_inject_fault: the nth prove_batch call device context `device` receives from now fails (0 disarms)"""
        self._check(self._L.masp_hip_ctx_inject_fault(self._h, int(device), int(nth)))

    def quotient_h(self, a, b, c, logm):
        a, b, c = _u8(a, 32), _u8(b, 32), _u8(c, 32)
        out = np.zeros(((1 << logm) - 1, 32), dtype=np.uint8)
        self._check(self._L.masp_hip_quotient_h(self._h, _p(a), _p(b), _p(c), a.shape[0], logm, _p(out)))
        return out

    def ntt(self, data, logm, inverse=False):
        d = _u8(data, 32).copy()
        self._check(self._L.masp_hip_ntt(self._h, _p(d), logm, 1 if inverse else 0))
        return d

    # ---- measurement hooks ----
    def batch_upload(self, jobs):
        jobs = list(jobs)
        arr = (JobStruct * len(jobs))()
        keep = []
        for i, job in enumerate(jobs):          # (slot, inputs, aux, r, s[, None[, aux_form]]): as marshal_jobs
            slot, inputs, aux, r, s = job[:5]
            arr[i], k = self._job(slot, inputs, aux, r, s, None, job[6] if len(job) > 6 else AUX_CANONICAL)
            keep.append(k)
        h = self._L.masp_hip_batch_upload(self._h, len(jobs), arr)
        if h < 0:
            raise MaspHipError(-h, self._L.masp_hip_last_error(self._h).decode(errors="replace"))
        return h, len(jobs)

    def batch_prove_resident(self, handle, n):
        out = np.zeros((n, 192), dtype=np.uint8)
        ms = C.c_float(0)
        self._check(self._L.masp_hip_batch_prove_resident(self._h, handle, _p(out), C.byref(ms)))
        return [out[i].tobytes() for i in range(n)], ms.value

    def batch_prove_resident_steps(self, handle, n, steps, rs=None):
        """Proves the n resident jobs `steps` times; rs: u8[steps, n, 64] (r | s per job and step) or None.
        -> (u8[steps, n, 192], HIP-event milliseconds)"""
        out = np.zeros((steps, n, 192), dtype=np.uint8)
        ms = C.c_float(0)
        if rs is not None:
            rs = np.ascontiguousarray(rs, dtype=np.uint8)
            assert rs.shape == (steps, n, 64)
        self._check(self._L.masp_hip_batch_prove_resident_steps(self._h, handle, steps, _p(rs), _p(out), C.byref(ms)))
        return out, ms.value

    def batch_free(self, handle):
        self._check(self._L.masp_hip_batch_free(self._h, handle))

    def profile_enable(self, on=True):
        self._check(self._L.masp_hip_profile_enable(self._h, 1 if on else 0))

    def profile_read_split(self):
        """-> summed ms of the profiled bucket stages by kernel group: plan / records / copies, pass 1, shared inversions, pass 2, XYZZ accumulation"""
        ms = (C.c_double * 8)()
        self._check(self._L.masp_hip_profile_read_split(self._h, ms))
        return dict(zip(("plan_records_copies", "k_tree_pass1", "k_binv", "k_tree_pass2", "k_msm_accumulate_pts"), [float(x) for x in ms[:5]]))

    LONE_MARKS = ("start", "msm_a", "s_A", "msm_b1", "r_B1", "msm_b2", "g_b", "quotient", "msm_h", "msm_l", "g_a_g_c", "complete")

    def profile_read_lone(self):
        """masp_hip_profile_read_lone: where the chains of the last lone proof ended (ms of GPU time, HIP events) -> dict by mark"""
        ms = (C.c_double * 12)()
        self._check(self._L.masp_hip_profile_read_lone(self._h, ms))
        return {k: round(float(v), 3) for k, v in zip(self.LONE_MARKS, ms)}

    def profile_read(self):
        """-> (summed k_msm_accumulate<G1> ms, launches, algorithmic bytes)"""
        t, l, b = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.masp_hip_profile_read(self._h, C.byref(t), C.byref(l), C.byref(b)))
        return t.value, l.value, b.value

    def sync(self):
        self._check(self._L.masp_hip_sync(self._h))

    def bench_msm(self, handle, job, which, iters):
        ms, nb = C.c_float(0), C.c_uint32(0)
        self._check(self._L.masp_hip_bench_msm(self._h, handle, job, which, iters, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value


class GpuVerifyingKey:
    """Groth16 batch verification with the Miller loops on the GPU (masp_hip_verify_batch): same interface as
    host.PreparedVerifyingKey.verify_batch = bellman `verify_proofs_batch` (sapling/verifier/batch.rs:24-31)."""

    def __init__(self, ctx, params):
        self._ctx = ctx
        buf = _u8(params)
        n_ic = int.from_bytes(buf[864:868].tobytes(), "big")
        self._buf = buf[:868 + 96 * n_ic].copy()
        self.n_public = n_ic - 1
        h = C.c_void_p()
        ctx._check(ctx._L.masp_hip_vk_prepare(ctx._h, _p(self._buf), self._buf.size, C.byref(h)))
        self._h = h

    def verify_batch(self, proofs, public_inputs, randomness=None):
        """proofs: list of 192-byte strings; public_inputs: one list of ints / 32-byte values per proof (excluding ONE).
        True iff all verify (up to 2^-127); False says at least one is invalid, not which."""
        import secrets
        n = len(proofs)
        if n == 0:
            return True
        if len(public_inputs) != n or any(len(pi) != self.n_public for pi in public_inputs):
            return False
        pr = np.frombuffer(b"".join(bytes(p) for p in proofs), dtype=np.uint8)
        if pr.size != 192 * n:
            return False
        pi = np.frombuffer(b"".join(_scalar32(x) for row in public_inputs for x in row), dtype=np.uint8) if self.n_public else None
        z = randomness if randomness is not None else secrets.token_bytes(16 * n)
        assert len(z) == 16 * n
        zb = np.frombuffer(z, dtype=np.uint8)
        ok = C.c_int(0)
        rc = self._ctx._L.masp_hip_verify_batch(self._ctx._h, self._h, n, _p(pr), _p(pi), self.n_public, _p(zb), C.byref(ok))
        if rc == 8:        # MASP_HIP_E_SCALAR_RANGE: a public input >= r — "does not verify", as host.PreparedVerifyingKey.verify_batch says
            return False
        self._ctx._check(rc)
        return ok.value == 1

    def close(self):
        if getattr(self, "_h", None) and getattr(self._ctx, "_h", None):
            self._ctx._L.masp_hip_vk_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
