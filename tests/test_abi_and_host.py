"""CPU-side tests: the C-ABI library loads and exports every symbol include/bosship.h declares
(no compute without a GPU), argument validation, host logic of the plugin mirror, golden fixtures
vs the oracle."""
import json
import math
import os
import re

import numpy as np
import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    entry.build()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "bosship.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(boss_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from boss_jl_amd import api
    lib = api.load_library()
    syms = header_symbols()
    assert len(syms) >= 21
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in bosship.h but not exported"
    assert sorted(api.SIGNATURES) == syms, "ctypes SIGNATURES must cover exactly the header's entry points"
    assert b"bosship" in lib.boss_version()


def test_no_gpu_fails_loudly_and_validates_arguments():
    import torch
    from boss_jl_amd import api
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert api.device_count() == 0
    with pytest.raises(api.BossError) as e:
        api.GP(np.zeros((2, 3)), np.zeros(3))
    assert e.value.code == api.BOSS_E_NO_DEVICE and "no CPU fallback" in str(e.value)
    # argument validation happens before any device work (gaussian_process.jl:227-233 asserts)
    with pytest.raises(api.BossError) as e:
        api.fit(np.zeros((2, 3)), np.zeros(3), "matern52", [-1.0, 1.0], 1.0, 1.0)
    assert e.value.code == api.BOSS_E_INVALID
    with pytest.raises(api.BossError) as e:
        api.fit(np.zeros((2, 3)), np.zeros(3), "matern52", [1.0], 1.0, 1.0)
    assert e.value.code == api.BOSS_E_INVALID
    # the size limit, the gradient model's x_dim limit and the new models' entry points without a device
    with pytest.raises(api.BossError) as e:
        api.GP(np.zeros((1, 46081)), np.zeros(46081))
    assert e.value.code == api.BOSS_E_INVALID and "46080" in str(e.value)
    with pytest.raises(api.BossError) as e:
        api.GradGP(np.zeros((17, 2)), np.zeros(2), np.zeros((17, 2)))
    assert e.value.code == api.BOSS_E_INVALID
    with pytest.raises(api.BossError) as e:
        api.GradGP(np.zeros((2, 3)), np.zeros(3), np.zeros((2, 3)))
    assert e.value.code == api.BOSS_E_NO_DEVICE
    with pytest.raises(api.BossError) as e:
        api.GibbsGP(np.zeros((2, 3)), np.zeros(3))
    assert e.value.code == api.BOSS_E_NO_DEVICE


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "boss.jl_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{fn} must not reference the oracle"
    for fn in os.listdir(os.path.join(pkg, "csrc")):
        assert "oracle" not in open(os.path.join(pkg, "csrc", fn)).read()


def test_shipped_library_has_no_work_skipping_switches():
    """BOSS_DBG (skip GEMM1 / GEMM2 / K* / the V store in the prediction kernel) and BOSS_EXP_NOREST (drop trailing
    updates) are timing experiments: they exist only in -DBOSS_EXPERIMENTS builds.  The shipped library must not even
    contain their names (a bench number must not be one environment variable away from skipping its FLOPs)."""
    from boss_jl_amd import api
    blob = open(api.LIB_PATH, "rb").read()
    for name in (b"BOSS_DBG", b"BOSS_EXP_NOREST"):
        assert name not in blob, name
    assert "BOSS_EXPERIMENTS" not in " ".join(entry.HIPCC_FLAGS)


def test_golden_fixtures_match_oracle():
    from oracle import gp_oracle as O
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "gp_golden.json")))
    assert len(cases) >= 8
    for c in cases:
        X, y, Xs = np.array(c["X"]), np.array(c["y"]), np.array(c["Xs"])
        post = O.gp_fit(X, y, c["kernel"], c["lengthscale"], c["amplitude"], c["noise_std"], mean=c["mean_X"],
                        discrete=c["discrete"])
        mu, var = O.gp_mean_and_var(post, Xs, c["mean_Xs"], clip=False)
        assert abs(post.logpdf - c["logpdf"]) <= 1e-11 * (1 + abs(c["logpdf"]))
        assert np.allclose(mu, c["mu"], rtol=0, atol=1e-11) and np.allclose(var, c["var"], rtol=0, atol=1e-11)
        acq = O.ei_acquisition([post], Xs, [1.0], [np.inf], c["best"], means_s=None if c["mean_Xs"] is None else [np.array(c["mean_Xs"])])
        assert np.allclose(acq, c["acq_ei"], rtol=0, atol=1e-12) and int(np.argmax(acq)) == c["argmax"]


# ---------------------------------------------------------------------------- host logic
def test_domain_and_best_so_far():
    import boss_jl_amd as B
    from boss_jl_amd.problem import best_so_far, in_bounds, in_domain
    dom = B.Domain(bounds=([5., 5.], [10., 10.]))
    X = np.array([[1., 5., 7., 10., 11.], [6., 6., 6., 6., 6.]])
    assert in_bounds(X, dom.bounds).tolist() == [False, True, True, True, False]
    dom2 = B.Domain(bounds=([0.], [10.]), discrete=[True], cons=lambda x: [x[0] - 2.0])
    assert in_domain(np.array([[1., 2., 2.5, 3.]]), dom2).tolist() == [False, True, False, True]
    # test/unit/test/acquisitions/expected_improvement.jl:165-179
    Y = np.array([[1., 2., 3.]])
    assert best_so_far(B.LinFitness([1.]), Y, [np.inf]) == 3.
    assert best_so_far(B.LinFitness([2.]), Y, [np.inf]) == 6.
    assert best_so_far(B.LinFitness([1.]), np.array([[10., 2., 3.]]), [5.]) == 3.
    assert best_so_far(B.LinFitness([1.]), Y, [0.]) is None
    assert best_so_far(B.LinFitness([1.]), np.zeros((1, 0)), [0.]) is None


def test_priors_and_sampler():
    import boss_jl_amd as B
    m = B.HipGaussianProcess(lengthscale_priors=[B.MvLogNormal([1., 1.], [1., 1.])] * 2,
                             amplitude_priors=[B.LogNormal()] * 2, noise_std_priors=[B.Dirac(1e-4)] * 2)
    p = m.params_sampler()(np.random.default_rng(0))
    assert p.lengthscales.shape == (2, 2) and p.amplitudes.shape == (2,) and np.all(p.noise_std == 1e-4)
    ll = m.params_loglike()
    assert math.isfinite(ll(p))
    p.noise_std[0] = 2e-4            # off the Dirac -> -Inf  (gaussian_process.jl test :318-356)
    assert ll(p) == -math.inf
    from scipy import stats
    assert abs(B.LogNormal().logpdf(1.7) - stats.lognorm(s=1.0).logpdf(1.7)) < 1e-12
    s = p.slice(1)
    assert s.lengthscales.shape == (2, 1)
    from boss_jl_amd.model import join_slices
    j = join_slices([p.slice(0), p.slice(1)])
    assert np.array_equal(j.lengthscales, p.lengthscales)


def test_shard_range_and_reduce_pairs():
    from boss_jl_amd.distributed import reduce_pairs, shard_range
    for n in (0, 1, 7, 8192, 8193):
        for w in (1, 2, 3, 8):
            cuts = [shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert reduce_pairs([(1.0, 5), (3.0, 9), (3.0, 2)]) == (3.0, 2)          # tie -> smaller index
    v, i = reduce_pairs([(1.0, 0), (float("nan"), 7), (5.0, 3)])
    assert i == 7 and v != v                                                   # NaN is the largest (Julia argmax)
    assert reduce_pairs([(-np.inf, 10), (-np.inf, 4)]) == (-np.inf, 4)


def test_asm_ring_kernels_do_not_spill(tmp_path):
    """The hand-counted prefetch ring (csrc/gemm_f64.hpp) is only correct if hipcc neither spills to
    AGPRs nor to scratch in the kernels that use it (it would copy ring registers before their loads
    have landed).  Checked on the code-object metadata, which cross-compiles without a GPU."""
    import subprocess
    out = tmp_path / "bosship.s"
    flags = [f for f in entry.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags +
                          ["-S", "--cuda-device-only", "-o", str(out), os.path.join(entry.CSRC, "bosship.hip")])
    txt = out.read_text()
    found = chain_found = 0
    for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        agpr, name, scratch, vgpr = int(m.group(1)), m.group(2), int(m.group(3)), int(m.group(4))
        if any(k in name for k in ("predict_kernel", "backsolve_kernel", "potrf_syrk", "potrf_colupd", "potrf_rowupd", "linv_level",
                                   "kinv_syrk", "linvt_kernel", "few_update", "few_back_update", "few_finish", "few_back_finish", "inv_fwd_kernel",
                                   "inv_bwd_kernel")):
            found += 1
            assert agpr == 0 and scratch == 0 and vgpr <= 256, (name, agpr, scratch, vgpr)
        # the resident chain's kernels sit on the critical path of every posterior update: nothing of theirs may go through scratch
        # memory (potrf_chain_kernel ran with 21 spilled registers and 16 scratch accesses per block until round 5); the rider's
        # step kernel uses the hand-counted ring as well
        if any(k in name for k in ("potrf_chain_kernel", "potrf_strips_kernel", "potrf_follow_kernel", "rider_step_kernel")):
            chain_found += 1
            assert scratch == 0, (name, scratch)
            if "rider_step_kernel" in name:
                assert agpr == 0 and vgpr <= 256, (name, agpr, vgpr)
    assert found >= 19 and chain_found == 4


def test_integration_doc_shows_the_shipped_julia_glue():
    """INTEGRATION.md embeds boss.jl_amd/julia/BOSSHip.jl verbatim (the binding a BOSS.jl maintainer adds); the glue declares
    its own parameter type — `update_parameters!` asserts typeof(problem.model) <: M for FittedParams{M}
    (/root/reference/src/types/problem.jl:177-179) — passes prior means to the acquisition and binds every entry point it calls."""
    jl = open(os.path.join(ROOT, "boss.jl_amd", "julia", "BOSSHip.jl")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert jl in doc
    assert "struct HipGPParams" in jl and "<: BOSS.ModelParams{HipGaussianProcess}" in jl
    assert "const HipGPParams = BOSS.GaussianProcessParams" not in jl
    assert "prior_means(posts, xs)" in jl and "finalizer(close, obj)" in jl
    syms = set(header_symbols())
    called = set(re.findall(r"\(:(boss_[a-z0-9_]+), lib\)", jl))
    assert called and called <= syms, called - syms
    for must in ("boss_gp_fit", "boss_gp_update", "boss_gp_predict", "boss_acq_ei", "boss_multi_acq_ei_cand", "boss_init", "boss_gp_loglike_batch",
                 "boss_gp_fit_batch", "boss_multi_acq_ei_outputs", "boss_multi_acq_ei_samples", "boss_multi_loglike_batch", "boss_multi_gp_update",
                 "boss_track_create", "boss_acq_ei_tracks", "boss_acq_ei_grad", "boss_gp_loglike_grad_batch"):
        assert must in called, must
    # EVERY entry point of the header is either bound by the glue or listed in it with the reason why not
    listed = dict(re.findall(r"# not bound: (boss_[a-z0-9_]+) — (.+)", jl))
    assert not (set(listed) & called), set(listed) & called
    missing = syms - called - set(listed)
    assert not missing, f"exported by include/bosship.h but neither bound in BOSSHip.jl nor listed as '# not bound: <symbol> — <reason>': {sorted(missing)}"
    assert set(listed) <= syms and all(len(r.strip()) >= 10 for r in listed.values())
    # the plugin types a BOSS.jl user switches to: the sampling trio AND the gradient-based pair (OptimizationAM / OptimizationMAP
    # semantics, /root/reference/src/acquisition_maximizers/optimization.jl:89-118, src/model_fitters/optimization.jl:146-164)
    for typ, sup in (("HipBatchAM", "BOSS.AcquisitionMaximizer"), ("HipSequentialBatchAM", "BOSS.AcquisitionMaximizer"),
                     ("HipGradientAM", "BOSS.AcquisitionMaximizer"), ("HipBatchedMAP", r"BOSS.ModelFitter\{BOSS.MAPParams\}"),
                     ("HipGradientMAP", r"BOSS.ModelFitter\{BOSS.MAPParams\}"), ("HipImportanceBI", r"BOSS.ModelFitter\{BOSS.BIParams\}")):
        assert re.search(r"struct %s\b[^\n]*<: %s" % (typ, sup), jl), typ
    for fn in ("maximize_acquisition(am::HipGradientAM", "estimate_parameters(fit::HipGradientMAP", "maximize_acquisition(sb::HipSequentialBatchAM",
               "estimate_parameters(f::HipImportanceBI", "estimate_parameters(f::HipBatchedMAP"):
        assert fn in jl, fn
    # importance resampling from the prior weights with the DATA likelihood alone (likelihood × prior would target likelihood × prior²)
    bi = jl[jl.index("function estimate_parameters(f::HipImportanceBI"):]
    bi = bi[:bi.index("\nend") + 4]
    assert "batched_data_loglike" in bi and "prior" not in bi.replace("the prior", "").replace("× prior", "")
    # every method of the model API (/root/reference/src/types/surrogate_model.jl:19-73) is defined for BOTH device models:
    # the plain GP (BASELINE configs 1-3, 5) and the semiparametric model (config 4), each with a parameter type of its own
    for model, params in (("HipGaussianProcess", "HipGPParams"), ("HipSemiparametric", "HipSemiparametricParams")):
        assert re.search(r"struct %s\b[^\n]*<: BOSS\.SurrogateModel" % model, jl), model
        assert re.search(r"struct %s\{.{0,300}?\}\s*<: BOSS\.ModelParams\{%s\}" % (params, model), jl, re.S), params
        for fn in ("model_posterior_slice", "data_loglike", "params_loglike", "_params_sampler", "vectorizer", "bijector", "make_discrete"):
            assert re.search(r"(function )?%s\(m::%s[,)]" % (fn, model), jl), (fn, model)
        assert re.search(r"sliceable\(::%s\)" % model, jl), model
        for fn in ("param_count", "param_lengths", "param_shapes"):
            assert re.search(r"BOSS\.%s\(p::%s\)" % (fn, params), jl), (fn, params)
    for fn in ("slice(m::HipGaussianProcess", "slice(p::HipGPParams", "join_slices(ps::AbstractVector{<:HipGPParams}"):   # sliceable model
        assert fn in jl, fn
    # the parametric mean reaches the device as mean vectors (semiparametric.jl:79-92), the BI flow as P×S handles (posterior.jl:15-19)
    assert "BOSS.add_mean(m.sp.nonparametric, m.sp.parametric(θ))" in jl and "batch_means(m::HipSemiparametric" in jl
    assert "BOSS.model_posterior(m::HipGaussianProcess, ps::AbstractVector{<:HipGPParams}" in jl and "BOSS.BIParams(samples" in jl
    mpb = jl[jl.index("function model_posteriors_batched"):]
    assert ":boss_gp_fit_batch" in mpb[:mpb.index("\nend")] and ":boss_gp_fit," not in mpb[:mpb.index("\nend")]   # ONE batched call per output, not S fits
    # every ccall passes as many arguments as its type tuple declares
    for m in re.finditer(r"ccall\(\(:(boss_[a-z0-9_]+), lib\), (\w+),\s*\(([^)]*)\)", jl, re.S):
        name, types = m.group(1), [t for t in m.group(3).replace("\n", " ").split(",") if t.strip()]
        hdr = re.search(r"\b" + name + r"\s*\(([^;]*?)\);", re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "bosship.h")).read(), flags=re.S), re.S)
        assert hdr, name
        nargs = 0 if hdr.group(1).strip() in ("", "void") else hdr.group(1).count(",") + 1
        assert len(types) == nargs, (name, len(types), nargs)


def test_column_update_takes_every_strip_exactly_once():
    """potrf_colupd_kernel applies `C -= P P^T` to 32×128 strips with a plain read-modify-write: two workgroups on one
    strip would subtract the panel twice.  The workgroup -> strip map (colupd_decode, shared by the kernel and this
    host walk) is checked for every launch form potrf_enqueue issues (host_factor.inc) — including the chain schedule's
    last tail step (m == 1, a grid of 5), where the critical-strip swap used to send all five workgroups to the δ^T strip."""
    import ctypes as C
    from boss_jl_amd import api
    lib = api.load_library()
    fn = lib.boss_debug_colupd_decode
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_int)] * 3
    BLK = 128

    def walk(G, k, m, ncols, jfirst, skipdiag, xblk, critical):
        R0, C0, cr = (C.c_int * G)(), (C.c_int * G)(), (C.c_int * G)()
        assert fn(G, k, m, ncols, jfirst, skipdiag, xblk, critical, R0, C0, cr) == 0
        return [(R0[t], C0[t], cr[t]) for t in range(G)]

    def expect(k, m, cols, skipdiag, xblk, skip_first_diag_of=None):
        want = set()
        for j in cols:
            for t in range(4 * (m - j)):
                want.add(((k + 1 + j) * BLK + 32 * t, (k + 1 + j) * BLK))
            want.add(((k + 1 + m) * BLK, (k + 1 + j) * BLK))              # δ^T strip of the column
        if skipdiag:
            for t in range(4):
                want.discard(((k + 1 + cols[0]) * BLK + 32 * t, (k + 1 + cols[0]) * BLK))
        if xblk >= 0:
            for t in range(4):
                want.add((xblk * BLK + 32 * t, xblk * BLK))
        return want

    def check(work, want, ncrit, first_eight_crit):
        done = [(r, c) for r, c, _ in work if r >= 0]
        assert len(done) == len(set(done)), "a strip is taken by more than one workgroup"
        assert set(done) == want
        assert sum(cr for r, _, cr in work if r >= 0) == ncrit
        if first_eight_crit:                                               # the critical strips are dispatched first
            assert all(cr == 1 and r >= 0 for r, _, cr in work[:8])

    for chain in (0, 1):
        for k in (0, 3, 10):
            for m in range(1, 14):
                # even step of the paired phase / plain look-ahead step: column k+1 (+ tile (k+2, k+2) under the chain schedule)
                xblk = k + 2 if (chain and m >= 2) else -1
                G = 4 * m + 1 + (4 if xblk >= 0 else 0)
                check(walk(G, k, m, 1, 0, chain, xblk, chain), expect(k, m, [0], chain, xblk), 8 if (chain and m >= 2) else 0, chain and m >= 2)
                # odd step: columns k+1, k+2 (+ tile (k+3, k+3))
                nc = 2 if m >= 2 else 1
                xblk = k + 3 if (chain and m - 2 > 0) else -1
                G = (4 * m + 1) + (4 * (m - 1) + 1 if nc == 2 else 0) + (4 if xblk >= 0 else 0)
                check(walk(G, k, m, nc, 0, chain, xblk, chain), expect(k, m, list(range(nc)), chain, xblk), 8 if (chain and m >= 2) else 0,
                      chain and m >= 2)
                # single-stream tail: the whole trailing triangle
                G = 2 * m * (m + 1) + m
                check(walk(G, k, m, m, 0, chain, -1, chain), expect(k, m, list(range(m)), chain, -1), 8 if (chain and m >= 2) else 0,
                      chain and m >= 2)
                # near part of a split bulk update: columns k+3, k+4 without (k+3, k+3), plus tile (k+5, k+5); never critical
                if m >= 5:
                    G = (4 * (m - 2) + 1) + (4 * (m - 3) + 1) + 4
                    check(walk(G, k, m, 2, 2, 1, k + 5, 0), expect(k, m, [2, 3], 1, k + 5), 0, False)


def test_chain_schedules_apply_every_panel_once_in_order():
    """The trailing-update schedules that run beside the resident chain (potrf_enqueue, BOSS_CHAIN_TRAIL = 0 and 1), replayed on the
    host through the workgroup -> strip map the kernel itself uses: every 32-row strip of every block column, and its δ^T row,
    receives every earlier panel exactly once and in ascending order (the last panel of a diagonal tile comes from the chain
    kernel) — which is also why all schedules give bit-identical factors — and every diagonal tile has all panels but its last one
    when the step before its own factorisation ends."""
    import ctypes as C
    from boss_jl_amd import api
    lib = api.load_library()
    dec = lib.boss_debug_colupd_decode
    dec.restype = C.c_int
    dec.argtypes = [C.c_int] * 8 + [C.POINTER(C.c_int)] * 3
    BLK = 128

    def colupd(applied, G, k, m, ncols, jfirst, npan, skipdiag, xblk, crit):
        R0, C0, cr = (C.c_int * G)(), (C.c_int * G)(), (C.c_int * G)()
        assert dec(G, k, m, ncols, jfirst, skipdiag, xblk, crit, R0, C0, cr) == 0
        seen = set()
        for t in range(G):
            if R0[t] < 0:
                continue
            key = (R0[t] // 32, C0[t] // BLK)
            assert key not in seen, "a strip is taken by two workgroups of one launch"
            seen.add(key)
            for p in range(k - npan + 1, k + 1):
                applied.setdefault(key, []).append(p)
        if m >= 2:
            assert sum(cr[t] for t in range(G) if R0[t] >= 0) == 8 and all(cr[t] for t in range(8))

    for mode in (0, 1):
        for nblk in range(3, 41):
            applied = {}

            def diag_ready(j, upto):                          # tile (j, j) carries the panels 0 .. upto-1
                for r in range(4 * j, 4 * j + 4):
                    assert applied.get((r, j), []) == list(range(upto)), (mode, nblk, j, applied.get((r, j)))

            k = 0
            if mode == 1:
                while k + 3 < nblk:
                    m = nblk - 1 - k
                    colupd(applied, 4 * m + 1 + 4, k, m, 1, 0, 1, 1, k + 2, 1)
                    diag_ready(k + 2, k + 1)                  # F(k+2) adds panel k+1 during step k+1
                    m = nblk - 2 - k
                    colupd(applied, 2 * m * (m + 1) + m, k + 1, m, m, 0, 2, 1, -1, 1)
                    diag_ready(k + 3, k + 2)
                    k += 2
            for kt in range(k, nblk - 1):
                m = nblk - 1 - kt
                colupd(applied, 2 * m * (m + 1) + m, kt, m, m, 0, 1, 1, -1, 1)
                if kt + 2 < nblk:
                    diag_ready(kt + 2, kt + 1)
            for j in range(1, nblk):                           # the chain kernel applies panel j-1 to tile (j, j)
                for r in range(4 * j, 4 * j + 4):
                    applied.setdefault((r, j), []).append(j - 1)
            for j in range(1, nblk):
                for r in list(range(4 * j, 4 * nblk)) + [4 * nblk]:
                    assert applied.get((r, j)) == list(range(j)), (mode, nblk, r, j, applied.get((r, j)))
            assert all(j >= 1 and r >= 4 * j for (r, j) in applied)
