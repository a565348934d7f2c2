"""Host-side mirror of the reference's proving API for the Groth16 hot path.

    masp_proofs::prover::LocalTxProver                      /root/reference/masp_proofs/src/prover.rs:27-33,55-95,156-261
    masp_proofs::sapling::prover::SaplingProvingContext     /root/reference/masp_proofs/src/sapling/prover.rs:26-275
    masp_primitives::sapling::prover::TxProver (trait)      /root/reference/masp_primitives/src/sapling/prover.rs:17-83

Same method names, argument meaning, return values and error behaviour; the difference is what happens inside
`create_random_proof`: witness synthesis runs in libmasp_host.so (C++), and NTT / MSM / assembly run on the MI355X in
libmasp_hip.so.  The reference is Rust; no Rust toolchain exists in the build image, so this layer is Python over the
two C ABIs (INTEGRATION.md shows the Rust FFI binding a maintainer would write instead).

Types cross this API in their canonical byte encodings: Jubjub scalars and bls12_381::Scalar as 32-byte little-endian
(ints are accepted too), Jubjub points as their 32-byte `to_bytes()` encoding, AssetType as its 32-byte identifier,
MerklePath as (siblings leaf-level-first, position).
"""
import os
import secrets
import threading

from . import host as H
from . import params as P
from .hip import CONVERT, OUTPUT, SPEND, Context

GROTH_PROOF_SIZE = 192  # masp_primitives/src/transaction/components.rs:15
FR = H.FR_MODULUS
RJ = H.JUBJUB_ORDER


class ProvingError(Exception):
    """The reference's `Err(())`."""


def _int(x):
    return x if isinstance(x, int) else int.from_bytes(bytes(x), "little")


class Rseed:
    """`Rseed` (masp_primitives/src/sapling.rs:643-647) with `Note::rcm` (:856-864): BeforeZip212 holds the note commitment randomness itself,
    AfterZip212 32 seed bytes from which it is derived — jubjub::Fr::from_bytes_wide(PRF^expand(rseed, [0x04])), PRF^expand = BLAKE2b-512
    personalised "MASP__ExpandSeed" (masp_primitives/src/keys.rs:5-20).  `spend_proof` takes one of these, or rcm as an int / 32 bytes."""

    def __init__(self, kind, value):
        assert kind in ("BeforeZip212", "AfterZip212")
        self.kind, self.value = kind, value

    @classmethod
    def before_zip212(cls, rcm):
        return cls("BeforeZip212", rcm)

    @classmethod
    def after_zip212(cls, rseed):
        assert len(bytes(rseed)) == 32
        return cls("AfterZip212", bytes(rseed))

    def rcm(self):
        if self.kind == "BeforeZip212":
            return self.value
        import hashlib
        h = hashlib.blake2b(digest_size=64, person=b"MASP__ExpandSeed")
        h.update(self.value)
        h.update(b"\x04")
        return int.from_bytes(h.digest(), "little") % RJ


class SaplingProvingContext:
    """bsk / cv_sum bookkeeping of one transaction (sapling/prover.rs:26-47, :69-75, :154, :177-183, :205, :228-234, :272).

    Independent of the proof bytes, so the proofs of one transaction may be produced in any order or in one batch."""

    def __init__(self):
        self.bsk = 0                                  # jubjub::Fr::zero()
        self.cv_sum = H.JUBJUB_IDENTITY               # jubjub::ExtendedPoint::identity()

    # The reference adds rcv to bsk BEFORE anything can fail and cv to cv_sum only after the proof verified
    # (sapling/prover.rs:69-75 vs :154): after an Err(()) the context holds the new bsk and the old cv_sum.
    def _bsk_add(self, rcv, subtract=False):
        self.bsk = (self.bsk - _int(rcv)) % RJ if subtract else (self.bsk + _int(rcv)) % RJ     # "Outputs subtract from the total."

    def _cv_add(self, cv, subtract=False):
        self.cv_sum = H.jubjub_add(self.cv_sum, cv, subtract=subtract)

    def _spend_like(self, rcv, cv):
        self._bsk_add(rcv)
        self._cv_add(cv)

    def _output(self, rcv, cv):
        self._bsk_add(rcv, subtract=True)
        self._cv_add(cv, subtract=True)

    def binding_sig(self, assets_and_values, sighash, rng=None):
        """= SaplingProvingContext::binding_sig (sapling/prover.rs:279-326).  assets_and_values: iterable of
        (asset identifier[32], value balance as a signed 128-bit int) — the components of the reference's `I128Sum`.
        -> 64-byte RedJubjub signature over bvk || sighash, or ProvingError (`Err(())`) if the value balances do not
        match the accumulated commitments or a balance is i128::MIN."""
        from . import redjubjub as RJS
        g_rcv = H.point_bytes(*H.generator_uv(3))            # value_commitment_randomness_generator()
        bvk = RJS.public_key(self.bsk, g_rcv)
        final_bvk = self.cv_sum
        for asset, value in assets_and_values:
            if not -(1 << 127) < value < (1 << 127):         # checked_abs fails for i128::MIN (sapling/mod.rs:14-20)
                raise ProvingError("bad value balance")
            # value_commitment_generator = the asset generator with its cofactor cleared (asset_type.rs), times |value|
            vb = H.jubjub_mul(H.jubjub_mul(H.asset_generator(asset), 8), abs(value))
            final_bvk = H.jubjub_add(final_bvk, vb, subtract=value >= 0)
        if bvk != final_bvk:
            raise ProvingError("value balance does not match the accumulated value commitments")
        msg = bvk + bytes(sighash)
        assert len(msg) == 64
        return RJS.sign(self.bsk, msg, g_rcv, **({"rng": rng} if rng else {}))


class _Permits:
    """A counting semaphore whose acquire(k) takes k permits at once or none (k sequential acquires of a plain semaphore from
    several threads can each end up holding a part of what they need), served in TICKET ORDER: the waiter at the head of the queue
    blocks later acquirers even if their smaller requests could be met — a 16-permit group of the oldest chunk can then not be
    overtaken for ever by later one-permit groups, which with all remaining permits held by later chunks left no chunk able to finish
    and release."""

    def __init__(self, n):
        self._n = n
        self._cv = threading.Condition()
        self._next_ticket = 0
        self._serving = 0
        self._abandoned = set()

    def _skip_abandoned(self):
        while self._serving in self._abandoned:
            self._abandoned.discard(self._serving)
            self._serving += 1

    def acquire(self, k=1):
        with self._cv:
            ticket = self._next_ticket
            self._next_ticket += 1
            try:
                while self._serving != ticket or self._n < k:
                    self._cv.wait()
            except BaseException:
                # a waiter interrupted in wait() (KeyboardInterrupt, an async exception) gives its ticket up: without this nobody would
                # ever serve it and every later acquirer would wait for ever
                self._abandoned.add(ticket)
                self._skip_abandoned()
                self._cv.notify_all()
                raise
            self._n -= k
            self._serving += 1
            self._skip_abandoned()
            self._cv.notify_all()

    def release(self, k=1):
        with self._cv:
            self._n += k
            self._cv.notify_all()


class LocalTxProver:
    """An implementation of `TxProver` using the MI355X prover.  Holds the three circuits' parameters for its lifetime."""

    def __init__(self, spend_params, output_params, convert_params, device=0, rng=None, self_verify=True, expected=P.EXPECTED, options=None,
                 context=None):
        """= LocalTxProver::from_bytes (prover.rs:81-95): parameter *bytes* in the bellman wire format, digests checked
        as `parse_parameters` does (lib.rs:333-388).  Malformed or mismatching parameters raise `params.ParameterError` /
        `hip.HipError` (the reference panics, lib.rs:290-293,337,359-362).  `expected=None`: parameters that are not the
        MPC files (benches, tests).  `context`: a `Context` the process already has for this device — the prover loads its circuits
        into it and leaves it open when it is closed itself (a process wants ONE context per device: its slots' scratch is sized once)."""
        spend_params, output_params, convert_params = P.parse_parameters(spend_params, output_params, convert_params, expected=expected)
        self._owns_ctx = context is None
        self._ctx = Context(device, **(options or {})) if context is None else context      # options: masp_hip_options fields (slots, batch_cap, ...)
        self._pool = {SPEND: [], OUTPUT: [], CONVERT: []}           # recycled page-locked aux buffers per circuit
        self._pool_lock = threading.Lock()
        self._rng = rng or (lambda: secrets.randbelow(FR))          # r, s <- OsRng (sapling/prover.rs:66,174,225)
        self._self_verify = self_verify
        # spend_vk / convert_vk: PreparedVerifyingKey (prover.rs:27-33, lib.rs:391-393); Output proofs are not self-checked
        self.spend_vk = H.PreparedVerifyingKey(spend_params)
        self.convert_vk = H.PreparedVerifyingKey(convert_params)
        # the batched self-checks of prove_batch run their Miller loops on the GPU (masp_hip_verify_batch); the single-proof
        # checks of the TxProver methods and the search for the culprit of a failing batch stay on the host
        self._gpu_vk = {"spend": self._ctx.prepare_verifying_key(spend_params), "convert": self._ctx.prepare_verifying_key(convert_params)}
        for slot, kind, params in ((SPEND, "spend", spend_params), (OUTPUT, "output", output_params), (CONVERT, "convert", convert_params)):
            cs, _ = H.circuit(kind)
            self._ctx.load_circuit(slot, params, cs)

    from_bytes = classmethod(lambda cls, spend, output, convert, **kw: cls(spend, output, convert, **kw))

    @classmethod
    def new(cls, spend_path, output_path, convert_path, **kw):
        """= LocalTxProver::new (prover.rs:55-64) = load_parameters (lib.rs:278-328): parameter files on disk, sizes
        checked before anything is read."""
        expected = kw.get("expected", P.EXPECTED)
        if expected is not None:
            for kind, path in zip(P.KINDS, (spend_path, output_path, convert_path)):
                P.verify_file_size(path, expected[kind].bytes, "masp " + kind)
        return cls(*(open(p, "rb").read() for p in (spend_path, output_path, convert_path)), **kw)

    @classmethod
    def with_default_location(cls, **kw):
        """= LocalTxProver::with_default_location (prover.rs:120-136): ~/.masp-params/masp-{spend,output,convert}.params"""
        d = P.default_params_folder()
        if d is None or not os.path.isdir(d):
            return None
        paths = [os.path.join(d, n) for n in (P.MASP_SPEND_NAME, P.MASP_OUTPUT_NAME, P.MASP_CONVERT_NAME)]
        if not all(os.path.exists(p) for p in paths):
            return None
        return cls.new(*paths, **kw)

    @classmethod
    def with_synthetic_parameters(cls, seed=0, device=0, **kw):
        """Parameters generated on the GPU from known toxic waste, as the reference's benches do with
        `generate_random_parameters` (benches/sapling.rs:24-36): the MPC parameter files cannot be downloaded here."""
        from .synthetic import toxic_waste
        ctx = Context(device)
        params = [ctx.generate_parameters(H.circuit(k)[0], toxic_waste(seed * 3 + i)) for i, k in enumerate(("spend", "output", "convert"))]
        ctx.close()
        p = cls(*params, device=device, expected=None, **kw)
        p.parameters = dict(zip(("spend", "output", "convert"), params))
        return p

    def close(self):
        for k in self._gpu_vk.values():
            k.close()
        if self._owns_ctx:
            self._ctx.close()

    def new_sapling_proving_context(self):
        return SaplingProvingContext()

    # ---- page-locked aux buffers: the synthesizer writes where the DMA engine reads (no staging copy of ~3 MB / Spend) ----
    def _aux_take(self, slot):
        with self._pool_lock:
            if self._pool[slot]:
                return self._pool[slot].pop()
        n_aux = H.circuit(("spend", "output", "convert")[slot])[0].n_aux
        return self._ctx.host_alloc(n_aux, 32)

    def _aux_reserve(self, slot, count, slab=32):
        """Top the pool of circuit `slot` up to `count` buffers, whole slabs per masp_hip_host_alloc, BEFORE the GPU gets busy:
        page-locking while batches are being proved waits on the runtime (measured 14-18 ms per 3.2 MB buffer from the
        synthesis threads, against 0.4 ms on an idle device)."""
        n_aux = H.circuit(("spend", "output", "convert")[slot])[0].n_aux
        with self._pool_lock:
            have = len(self._pool[slot])
        while have < count:
            k = min(slab, count - have)
            big = self._ctx.host_alloc(n_aux * k, 32)
            with self._pool_lock:
                self._pool[slot].extend(big[i * n_aux:(i + 1) * n_aux] for i in range(k))
            have += k

    def _aux_give(self, jobs):
        """Return the aux buffers of finished jobs to the pool (the job dicts must not be proved again afterwards)."""
        with self._pool_lock:
            for j in jobs:
                if j is None:            # a synthesis task that returned at once because the call had already failed
                    continue
                buf = j.pop("_pinned", None)
                if buf is not None:
                    self._pool[j["slot"]].append(buf)

    # ---- witness preparation (host) and proving (GPU) are split so that batches can be formed ----
    def prepare_spend(self, proof_generation_key, diversifier, rcm, ar, asset_type, value, anchor, merkle_path, rcv):
        ak, nsk = proof_generation_key
        siblings, position = merkle_path
        buf = self._aux_take(SPEND)
        try:
            inputs, aux, cv, rk, nf = H.spend_assignment(ak, nsk, diversifier, rcm, ar, asset_type, value, anchor, siblings, position, rcv,
                                                         aux_out=buf, montgomery=True)
        except H.HostError as e:
            self._aux_give([dict(slot=SPEND, _pinned=buf)])
            raise ProvingError(str(e)) from None           # invalid diversifier -> Err(()) (sapling/prover.rs:84)
        return dict(slot=SPEND, inputs=inputs, aux=aux, aux_form=1, cv=cv, rk=rk, nf=nf, rcv=rcv, _pinned=buf)

    def prepare_output(self, esk, payment_address, rcm, asset_type, value, rcv):
        diversifier, pk_d = payment_address
        buf = self._aux_take(OUTPUT)
        try:
            inputs, aux, cv = H.output_assignment(esk, diversifier, pk_d, rcm, asset_type, value, rcv, aux_out=buf, montgomery=True)
        except H.HostError as e:
            self._aux_give([dict(slot=OUTPUT, _pinned=buf)])
            raise ProvingError(str(e)) from None
        return dict(slot=OUTPUT, inputs=inputs, aux=aux, aux_form=1, cv=cv, rcv=rcv, _pinned=buf)

    def prepare_convert(self, allowed_conversion, value, anchor, merkle_path, rcv):
        """allowed_conversion: a host.AllowedConversion (masp_primitives/src/convert.rs:22-29), or just its generator point
        (32 bytes) — the only part of it the circuit sees."""
        siblings, position = merkle_path
        generator = allowed_conversion.generator if isinstance(allowed_conversion, H.AllowedConversion) else allowed_conversion
        buf = self._aux_take(CONVERT)
        try:
            inputs, aux, cv = H.convert_assignment(generator, value, anchor, siblings, position, rcv, aux_out=buf, montgomery=True)
        except H.HostError as e:
            self._aux_give([dict(slot=CONVERT, _pinned=buf)])
            raise ProvingError(str(e)) from None
        return dict(slot=CONVERT, inputs=inputs, aux=aux, aux_form=1, cv=cv, rcv=rcv, _pinned=buf)

    def prepare_group(self, kind, kws):
        """prepare_<kind> for several descriptions of ONE circuit in a single native call (host.GROUP at a time): the Merkle blocks of
        Spend / Convert witnesses are synthesised side by side (csrc/host/circuits.h merkle_block_batch), ~2x the witnesses per
        second and thread.  -> list of job dicts, a ProvingError in the place of a description that cannot be proved."""
        if kind == "output":
            out = []
            for kw in kws:
                try:
                    out.append(self.prepare_output(**kw))
                except ProvingError as e:
                    out.append(e)
            return out
        slot = SPEND if kind == "spend" else CONVERT
        bufs = [self._aux_take(slot) for _ in kws]
        if kind == "spend":
            items = [(*kw["proof_generation_key"], kw["diversifier"], kw["rcm"], kw["ar"], kw["asset_type"], kw["value"], kw["anchor"],
                      *kw["merkle_path"], kw["rcv"]) for kw in kws]
            res = H.spend_assignments(items, aux_outs=bufs, montgomery=True)
        else:
            gens = [kw["allowed_conversion"].generator if isinstance(kw["allowed_conversion"], H.AllowedConversion) else kw["allowed_conversion"] for kw in kws]
            items = [(g, kw["value"], kw["anchor"], *kw["merkle_path"], kw["rcv"]) for g, kw in zip(gens, kws)]
            res = H.convert_assignments(items, aux_outs=bufs, montgomery=True)
        out = []
        for kw, buf, r in zip(kws, bufs, res):
            if isinstance(r, Exception):
                self._aux_give([dict(slot=slot, _pinned=buf)])
                out.append(ProvingError(str(r)))              # invalid diversifier -> Err(()) (sapling/prover.rs:84)
            elif kind == "spend":
                out.append(dict(slot=SPEND, inputs=r[0], aux=r[1], aux_form=1, cv=r[2], rk=r[3], nf=r[4], rcv=kw["rcv"], _pinned=buf))
            else:
                out.append(dict(slot=CONVERT, inputs=r[0], aux=r[1], aux_form=1, cv=r[2], rcv=kw["rcv"], _pinned=buf))
        return out

    def prove_batch(self, ctx, descriptions, threads=None, rs=None, chunk=None, progress=None, in_flight=None):
        """Batched form of the serial per-description loops of `SaplingBuilder::build`
        (/root/reference/masp_primitives/src/transaction/components/sapling/builder.rs:935-1140):
        descriptions = [("spend", kwargs) | ("output", kwargs) | ("convert", kwargs)] with the keyword arguments of
        prepare_spend / prepare_output / prepare_convert.

        Three stages run as a pipeline over chunks of `chunk` descriptions (default: n / 8 within 64 .. 256):
        witness synthesis on `threads` host threads (the C++ synthesizer releases the GIL), proving on the GPU (each
        chunk in flight owns one slot of the native context, which is re-entrant), and self-verification of the Spend /
        Convert proofs of a finished chunk as one `verify_proofs_batch`-style check per circuit, Miller loops on the GPU
        (masp_hip_verify_batch; a failing batch is re-checked proof by proof on the host, so the outcome is that of the
        per-proof checks at sapling/prover.rs:148,266).
        The context accumulates in description order afterwards, so bsk / cv_sum end up exactly as in the serial loops.
        `progress(done, total)` mirrors the builder's `Progress` notifications (builder.rs:946-952 etc.).
        -> list of (zkproof, cv[, rk])"""
        from concurrent.futures import ThreadPoolExecutor
        self._warm_wait()
        n = len(descriptions)
        if rs is None:
            rs = [(self._rng(), self._rng()) for _ in range(n)]
        threads = threads or H.effective_cpus()
        # descriptions per GPU call: an eighth of the list, between 64 and the launch-sequence size (256).  Short lists want
        # the first call to start early (1 024 Spends: 769 proofs/s in chunks of 128 against 730 in chunks of 256), long mixed
        # lists want full batches per circuit (4 096 mixed: 1 104 proofs/s in chunks of 256 against 999 in chunks of 128)
        # (with the lockstep synthesizer the host is no longer what a chunk waits for: full launch sequences from 4 x 256 descriptions on)
        cap = self._ctx.options["batch_cap"]
        chunk = chunk or (cap if n >= 4 * cap else max(64, min(cap, n // 8)))
        # GPU calls in flight: one per slot of the native context and one more, so that a slot is taken again at once while the
        # chunk that held it is being self-verified (the call waits inside the library for a free slot)
        in_flight = in_flight or max(1, self._ctx.options["slots"]) + 1
        prep = {"spend": self.prepare_spend, "output": self.prepare_output, "convert": self.prepare_convert}
        done = [0]

        def public_input(kind, kw, job):
            if kind == "spend":       # sapling/prover.rs:121-145
                return list(H.point_uv(job["rk"])) + list(H.point_uv(job["cv"])) + [_int(kw["anchor"])] + H.multipack(job["nf"])
            return list(H.point_uv(job["cv"])) + [_int(kw["anchor"])]          # convert: sapling/prover.rs:256-263

        # synthesis may run ahead of the GPU only so far: every job in flight owns a page-locked aux buffer (3.2 MB / Spend)
        window = (in_flight + 1) * chunk + threads * H.GROUP  # synthesis runs one chunk ahead (measured: two buy nothing)
        ahead = _Permits(window)
        # ... per circuit: the most descriptions of that circuit any `window` consecutive ones hold
        most, inside = {}, {}
        for i, (kind, _) in enumerate(descriptions):
            inside[kind] = inside.get(kind, 0) + 1
            if i >= window:
                inside[descriptions[i - window][0]] -= 1
            if inside[kind] > most.get(kind, 0):
                most[kind] = inside[kind]
        for kind, slot in (("spend", SPEND), ("output", OUTPUT), ("convert", CONVERT)):
            self._aux_reserve(slot, most.get(kind, 0))

        abort = threading.Event()              # set when a chunk fails: queued synthesis tasks then return at once

        # synthesis tasks: runs of one circuit, host.GROUP descriptions at most, never across a chunk boundary — one native call each
        groups, i = [], 0
        while i < n:
            kind, j = descriptions[i][0], i + 1
            limit = min(n, (i // chunk + 1) * chunk, i + H.GROUP)
            while j < limit and descriptions[j][0] == kind:
                j += 1
            groups.append((i, j))
            i = j

        def synthesize(lo, hi):
            ahead.acquire(hi - lo)
            if abort.is_set():
                return [None] * (hi - lo)
            return self.prepare_group(descriptions[lo][0], [kw for _, kw in descriptions[lo:hi]])

        failed = None
        done_lock = threading.Lock()
        with ThreadPoolExecutor(max_workers=threads) as synth, ThreadPoolExecutor(max_workers=in_flight) as gpu:
            group_futures = [synth.submit(synthesize, lo, hi) for lo, hi in groups]
            where = [None] * n                        # description k -> (its group's future, its index there)
            for f, (lo, hi) in zip(group_futures, groups):
                for k in range(lo, hi):
                    where[k] = (f, k - lo)

            def run_chunk(lo):
                hi = min(n, lo + chunk)
                if abort.is_set():                      # an earlier chunk failed: the transaction is lost, do not spend GPU time on it
                    raise ProvingError("batch aborted")
                jobs = [where[k][0].result()[where[k][1]] for k in range(lo, hi)]
                errors = [j for j in jobs if isinstance(j, Exception)]
                if errors or any(j is None for j in jobs):     # a description that cannot be proved, or synthesis saw the abort flag
                    self._aux_give([j for j in jobs if isinstance(j, dict)])
                    for j in jobs:
                        if isinstance(j, dict):
                            j["_given"] = True
                    raise errors[0] if errors else ProvingError("batch aborted")
                try:
                    proofs = self.prove_prepared(jobs, rs[lo:hi])
                    if self._self_verify:
                        for kind, vk in (("spend", self.spend_vk), ("convert", self.convert_vk)):
                            sel = [i for i in range(hi - lo) if descriptions[lo + i][0] == kind]
                            if not sel:
                                continue
                            pis = [public_input(kind, descriptions[lo + i][1], jobs[i]) for i in sel]
                            if not self._gpu_vk[kind].verify_batch([proofs[i] for i in sel], pis):
                                bad = [lo + i for i, pi in zip(sel, pis) if not vk.verify(proofs[i], pi)]
                                raise ProvingError("proof(s) %s failed self-verification" % bad)
                finally:
                    self._aux_give(jobs)                   # proved and checked, or failed: the aux buffers go back to the pool
                    for j in jobs:
                        j["_given"] = True
                ahead.release(hi - lo)
                with done_lock:
                    done[0] += hi - lo
                    so_far = done[0]
                if progress is not None:
                    progress(so_far, n)
                # this chunk's share of cv_sum (outputs subtract), off the caller's thread: decompressing 2 048 commitments one by
                # one afterwards was 6 % of a call
                cv_part = H.jubjub_sum([j["cv"] for j in jobs], [descriptions[lo + i][0] == "output" for i in range(hi - lo)])
                return jobs, proofs, cv_part
            first_error = []

            def guarded(lo):
                try:
                    return run_chunk(lo)
                except BaseException as e:          # the first failure stops the chunks still queued (they check `abort`)
                    with done_lock:
                        if not abort.is_set():
                            first_error.append(e)   # the cause, not the "batch aborted" of the chunks that follow it
                            abort.set()
                            ahead.release(n + window)
                    raise
            try:
                results = list(gpu.map(guarded, range(0, n, chunk)))
            except BaseException as e:             # (an invalid diversifier, a failed self-check, a device error ...)
                failed = first_error[0] if first_error else e
        if failed is not None:
            # both executors have shut down, so every synthesis task has finished: whatever it produced and no chunk gave
            # back (the failing chunk's earlier jobs, later chunks, tasks that ran past the abort check) returns its
            # page-locked buffer now — a service that keeps hitting bad inputs must not accumulate pinned memory.
            # The context is left untouched: the builder drops it together with the failed transaction.
            left = [j for f in group_futures if f.done() and not f.cancelled() and f.exception() is None for j in f.result()
                    if isinstance(j, dict) and not j.get("_given")]
            self._aux_give(left)
            raise failed
        jobs = [j for js, _, _ in results for j in js]
        proofs = [p for _, ps, _ in results for p in ps]
        out = []
        for (kind, kw), job, zk in zip(descriptions, jobs, proofs):
            ctx._bsk_add(kw["rcv"], subtract=kind == "output")
            out.append((zk, job["cv"], job["rk"]) if kind == "spend" else (zk, job["cv"]))
        ctx.cv_sum = H.jubjub_sum([part for _, _, part in results], acc=ctx.cv_sum)      # (an abelian group: the chunks' sums in any order)
        return out

    def warm_up(self, spends=None, outputs=0, converts=0, threads=None, background=False):
        """Pay at load time what the first prove_batch of a fresh prover otherwise pays in its own latency (round 4: 3.4 s for the first call
        over 4 096 descriptions against 2.5 s warm): (1) the page-locked aux pool — as many buffers per circuit as a prove_batch over
        `spends` / `outputs` / `converts` descriptions keeps in flight (None: a long list; page-locking 3.2 MB takes 0.4 ms on an idle
        device and 15 ms next to running batches); (2) every slot's device scratch at its final size — one launch sequence of
        `batch_cap` proofs per slot and circuit over an all-zero witness (the scratch of a sequence does not depend on the witness: the
        tree arena is sized for full-width scalars), `slots` calls side by side so that each lands on a slot of its own.
        background=True: on a thread; the next proving call waits for it.  = what `LocalTxProver::new` (prover.rs:55-95) has no
        counterpart for: bellperson allocates per proof."""
        if background:
            self._warm_error = None

            def run():
                try:
                    self.warm_up(spends, outputs, converts, threads)
                except BaseException as e:      # kept for the next proving call: a daemon thread's exception is otherwise lost
                    self._warm_error = e
            t = threading.Thread(target=run, daemon=True)
            self._warm = t
            t.start()
            return
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        threads = threads or H.effective_cpus()
        cap, slots = self._ctx.options["batch_cap"], max(1, self._ctx.options["slots"])
        window = (slots + 2) * cap + threads * H.GROUP            # prove_batch's own bound for a long list (in_flight = slots + 1)
        for slot, kind, want in ((SPEND, "spend", spends), (OUTPUT, "output", outputs), (CONVERT, "convert", converts)):
            want = window if want is None else min(int(want), window)
            if want <= 0:
                continue
            self._aux_reserve(slot, want)
            cs, _ = H.circuit(kind)
            buf = self._aux_take(slot)
            buf[:] = 0                                            # Montgomery zero = canonical zero: a witness of zeros
            inputs = np.zeros((cs.n_inputs, 32), np.uint8)
            inputs[0, 0] = 1                                      # the constant ONE
            n = min(cap, max(want, 1))
            jobs = [(slot, inputs, buf, 1 + k, 2 + k, None, 1) for k in range(n)]
            with ThreadPoolExecutor(slots) as ex:
                list(ex.map(lambda _: self._ctx.prove_batch(jobs), range(slots)))
            self._aux_give([dict(slot=slot, _pinned=buf)])

    def _warm_wait(self):
        t = getattr(self, "_warm", None)
        if t is not None and t is not threading.current_thread():
            t.join()
            self._warm = None
            e, self._warm_error = getattr(self, "_warm_error", None), None
            if e is not None:           # out of memory while sizing the slots, a failed launch sequence: the caller must not prove on
                raise RuntimeError("LocalTxProver.warm_up(background=True) failed: %r" % (e,)) from e   # a context in an unknown state

    def prove_prepared(self, jobs, rs=None):
        """jobs: outputs of prepare_*; rs: optional explicit [(r, s)] (deterministic replay) -> list of 192-byte proofs."""
        if rs is None:
            rs = [(self._rng(), self._rng()) for _ in jobs]
        self._warm_wait()
        # (the aux assignments come from libmasp_host as Montgomery residues — masp_hip_job::aux_form = 1: no conversion on the host)
        return self._ctx.prove_batch([(j["slot"], j["inputs"], j["aux"], r, s, None, j.get("aux_form", 0)) for j, (r, s) in zip(jobs, rs)])

    # ---- the TxProver methods ----
    def spend_proof(self, ctx, proof_generation_key, diversifier, rseed, ar, asset_type, value, anchor, merkle_path, rcv, rs=None):
        """-> (zkproof[192], cv, rk).  `rseed`: an `Rseed`, or the note commitment randomness rcm = note.rcm() itself (the BeforeZip212 form)."""
        if isinstance(rseed, Rseed):
            rseed = rseed.rcm()                                                # note.rcm() (sapling/prover.rs:95-101 builds the note from rseed)
        ctx._bsk_add(rcv)                                                      # :69-75, before anything can fail
        job = self.prepare_spend(proof_generation_key, diversifier, rseed, ar, asset_type, value, anchor, merkle_path, rcv)
        try:
            zkproof = self.prove_prepared([job], None if rs is None else [rs])[0]
        finally:
            self._aux_give([job])
        if self._self_verify:
            # public input built from the natively computed rk, cv, anchor and nullifier (sapling/prover.rs:121-145)
            public_input = list(H.point_uv(job["rk"])) + list(H.point_uv(job["cv"])) + [_int(anchor)] + H.multipack(job["nf"])
            if not self.spend_vk.verify(zkproof, public_input):
                raise ProvingError("spend proof failed self-verification")      # .map_err(|_| ())? at :148
        ctx._cv_add(job["cv"])                                                 # :154
        return zkproof, job["cv"], job["rk"]

    def output_proof(self, ctx, esk, payment_address, rcm, asset_type, value, rcv, rs=None):
        """-> (zkproof[192], cv); infallible for valid inputs like the reference (it panics if proving fails)."""
        ctx._bsk_add(rcv, subtract=True)                                       # :177-183
        job = self.prepare_output(esk, payment_address, rcm, asset_type, value, rcv)
        try:
            zkproof = self.prove_prepared([job], None if rs is None else [rs])[0]
        finally:
            self._aux_give([job])
        ctx._cv_add(job["cv"], subtract=True)                                  # :205
        return zkproof, job["cv"]

    def convert_proof(self, ctx, allowed_conversion, value, anchor, merkle_path, rcv, rs=None):
        """-> (zkproof[192], cv)"""
        ctx._bsk_add(rcv)                                                      # :228-234
        job = self.prepare_convert(allowed_conversion, value, anchor, merkle_path, rcv)
        try:
            zkproof = self.prove_prepared([job], None if rs is None else [rs])[0]
        finally:
            self._aux_give([job])
        if self._self_verify:
            public_input = list(H.point_uv(job["cv"])) + [_int(anchor)]            # sapling/prover.rs:256-263
            if not self.convert_vk.verify(zkproof, public_input):
                raise ProvingError("convert proof failed self-verification")    # :266
        ctx._cv_add(job["cv"])                                                 # :272
        return zkproof, job["cv"]

    def binding_sig(self, ctx, assets_and_values, sighash):
        return ctx.binding_sig(assets_and_values, sighash)
