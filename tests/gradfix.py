"""Full-size parity against COMMITTED fixtures (tests/golden/full/*.npz) instead of a CPU-oracle run on the GPU box.

tests/golden/make_golden_fullsize.py evaluates the pinned oracle (oracle/*.py, itself tied to the imported reference by
tests/golden/make_golden.py + tests/test_oracle_golden.py) in the build container, in fp32 AND fp64 plus the two
kink-shifted fp64 passes, and stores for every case

  * the forward outputs as spatially strided samples (every 3rd row / column) + (sum, |sum|, sum of squares) checksums of
    the whole tensor, in both precisions,
  * every logged scalar in both precisions,
  * for EVERY gradient tensor: every N-th element (N = the smallest prime >= numel / SAMPLES, 1 for small tensors, flattened
    logical OIHW order) of the fp32 and of the fp64 oracle gradient, the whole-tensor maxima s32 / s64, the fp32 oracle's own
    distance from fp64 (`own`), the kink bracket (`kink`) and whole-tensor checksums of the fp64 gradient,
  * a checksum of the inputs / weights the numbers belong to (so a drift of the procedural generators is named as such).

The `-m gpu` tests run the HIP path on the same seeded inputs and apply the SAME acceptance rules as before (rule "tryon":
tests/test_parity_bs4_gpu.py r03; rule "sams": tests/test_sams_gpu.py r03) to the sampled elements, plus a whole-tensor
energy check.  Every tensor's ROUTE through the rule (fp32 / fp64 / scalar / kink / zero ...) is tallied; the tally of a
GPU run is committed (tests/golden/full/routes.json) and a tensor that needs a WEAKER route than recorded fails the test.
"""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = os.path.join(HERE, "golden", "full")
ROUTES_FILE = os.path.join(FULL, "routes.json")
SAMPLES = 2048
GRAD_REL = 2e-3
GRAD_FLOOR = 2e-7

# weakest last; a tensor may move LEFT between rounds, never right
ROUTE_ORDER = {"tryon": ["zero", "fp32", "fp64", "fp64-scalar", "fp64+kink"],
               "sams": ["zero", "base", "own", "kink"]}


def _is_prime(n):
    if n < 2:
        return False
    i = 2
    while i * i <= n:
        if n % i == 0:
            return False
        i += 1
    return True


def sample_stride(numel, samples=SAMPLES):
    if numel <= samples:
        return 1
    n = -(-numel // samples)
    while not _is_prime(n):
        n += 1
    return n


def checksums(t):
    t = t.detach().double().reshape(-1)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


# ------------------------------------------------------------------------------------------------
# writing (build container only)
# ------------------------------------------------------------------------------------------------
def pack_grads(out, prefix, g32, g64, kink=None, samples=SAMPLES):
    """g32 / g64: {name: tensor} of the fp32 / fp64 oracle gradients (same keys); kink: {name: spread} or None."""
    names = sorted(g32)
    assert names == sorted(g64), set(g32) ^ set(g64)
    stats = np.zeros((len(names), 8), np.float64)
    numel, stride, offs, s32, s64 = [], [], [0], [], []
    for i, k in enumerate(names):
        a, b = g32[k].detach().contiguous().reshape(-1), g64[k].detach().contiguous().reshape(-1).double()
        assert a.shape == b.shape, k
        n = a.numel()
        st = sample_stride(n, samples)
        numel.append(n)
        stride.append(st)
        s32.append(a[::st].float().numpy().copy())
        s64.append((b[::st] - a[::st].double()).float().numpy().copy())  # fp64 value = fp32 sample + this difference
        offs.append(offs[-1] + s32[-1].size)
        cs = checksums(b)
        stats[i] = [a.abs().max().item(), b.abs().max().item(), (a.double() - b).abs().max().item(),
                    float((kink or {}).get(k, 0.0)), cs[0], cs[1], cs[2], checksums(a)[2]]
    out[prefix + "names"] = np.array(names)
    out[prefix + "shapes"] = np.array([str(tuple(g32[k].shape)) for k in names])
    out[prefix + "numel"] = np.array(numel, np.int64)
    out[prefix + "stride"] = np.array(stride, np.int64)
    out[prefix + "offs"] = np.array(offs, np.int64)
    out[prefix + "stats"] = stats
    out[prefix + "g32"] = np.concatenate(s32) if s32 else np.zeros(0, np.float32)
    out[prefix + "d64"] = np.concatenate(s64) if s64 else np.zeros(0, np.float32)


def pack_output(out, key, t32, t64=None, stride=3):
    """A forward output: spatially strided samples + whole-tensor checksums, fp32 (and fp64) oracle values."""
    out[key + ":s32"] = t32.detach()[..., ::stride, ::stride].contiguous().float().numpy()
    out[key + ":cs32"] = checksums(t32)
    out[key + ":stride"] = np.int64(stride)
    out[key + ":shape"] = np.array(t32.shape, np.int64)
    if t64 is not None:
        d = t64.detach().double() - t32.detach().double()  # fp64 value = fp32 sample + this difference (exact to ~1e-13)
        out[key + ":d64"] = d[..., ::stride, ::stride].contiguous().float().numpy()
        out[key + ":max64"] = np.float64(t64.detach().abs().max().item())
        out[key + ":cs64"] = checksums(t64)
        out[key + ":referr"] = np.float64((t32.detach().double() - t64.detach().double()).abs().max().item())


def input_digest(tensors):
    """Order-independent checksum of a dict of tensors (weights / batch): the fixture belongs to exactly these inputs."""
    acc = np.zeros(3, np.float64)
    for k in sorted(tensors):
        v = tensors[k]
        if torch.is_tensor(v) and v.is_floating_point():
            acc += checksums(v)
    return acc


def save(name, out):
    os.makedirs(FULL, exist_ok=True)
    path = os.path.join(FULL, name + ".npz")
    np.savez_compressed(path, **out)
    return path


# ------------------------------------------------------------------------------------------------
# reading (GPU tests)
# ------------------------------------------------------------------------------------------------
def load(name):
    path = os.path.join(FULL, name + ".npz")
    assert os.path.exists(path), (f"{path} is missing: regenerate it in the build container with "
                                  f"python tests/golden/make_golden_fullsize.py {name}")
    return np.load(path, allow_pickle=False)


def check_digest(fix, key, tensors):
    mine, ref = input_digest(tensors), fix[key]
    assert np.allclose(mine, ref, rtol=1e-12, atol=1e-9), (
        f"the fixture's {key} was generated for other inputs than the test just built ({mine} vs {ref}): the procedural "
        f"generator changed - regenerate tests/golden/full with make_golden_fullsize.py")


def _nchw_cpu(t):
    return t.detach().cpu()  # logical NCHW order whatever the device pitch (NHWC rows) is


def check_output(fix, key, ours, atol, what, rel_to_max=False, either=True, mode=None):
    """Sampled elements against the oracle values.  mode "fp32": within atol of the fp32 oracle (the north-star statement);
    "either": within atol of the fp32 oracle or no further from the fp64 value than atol + the fp32 oracle's own distance
    from it (the r03 `assert_close_either` rule, C5); "min": the whole sample within atol of the fp32 oracle or the whole
    sample within atol of the fp64 one (the r03 SAMS rule).  Then whole-tensor checksums within the bound per-element
    agreement implies."""
    mode = mode or ("either" if either else "fp32")
    st = int(fix[key + ":stride"])
    o_full = _nchw_cpu(ours).double()
    assert tuple(o_full.shape) == tuple(int(x) for x in fix[key + ":shape"]), (what, key, tuple(o_full.shape))
    o = o_full[..., ::st, ::st]
    r32 = torch.from_numpy(fix[key + ":s32"]).double()
    scale = 1.0
    has64 = (key + ":d64") in fix.files
    if rel_to_max:
        scale = max(1.0, float(fix[key + ":max64"]) if has64 else float(np.abs(fix[key + ":s32"]).max()))
    d32 = (o - r32).abs()
    tol = atol * scale
    msg = f"max |ours - fp32 oracle| {float(d32.max()):.3e}"
    if has64:
        r64 = r32 + torch.from_numpy(fix[key + ":d64"]).double()
        referr = float(fix[key + ":referr"])
        d64 = (o - r64).abs()
        msg += f", max |ours - fp64| {float(d64.max()):.3e}, max |fp32 oracle - fp64| {referr:.3e}"
    if mode == "either" and has64:
        bad = (d32 > tol) & (d64 > tol + referr)
        assert float(d64.max()) <= 2 * max(referr, tol / 2), f"{what} {key}: further from the exact value than twice the reference is; {msg}"
    elif mode == "min" and has64:
        bad = d32 > tol if float(d32.max()) <= float(d64.max()) else d64 > tol
    else:
        bad = d32 > tol
    print(f"[{what}] {key}: {msg} ({o.numel()} of {o_full.numel()} elements sampled, atol {tol:g}, rule {mode})")
    assert not bad.any(), f"{what} {key}: {int(bad.sum())}/{bad.numel()} sampled elements out of tolerance; {msg}"
    # whole tensor: |sum(ours) - sum(ref)| <= numel * tol and the matching bound on the sum of squares are implied by
    # element-wise agreement; a garbage region off the sample lattice (one 64 x 64 tile of O(1) errors) breaks both
    cs = checksums(o_full)
    n = o_full.numel()
    for ref_key, extra in ((":cs32", 0.0), (":cs64", float(fix[key + ":referr"]) if has64 else 0.0)):
        if key + ref_key not in fix.files:
            continue
        ref_cs, t = fix[key + ref_key], tol + extra
        if abs(cs[0] - ref_cs[0]) <= n * t and abs(cs[2] - ref_cs[2]) <= 2 * (ref_cs[2] * n) ** 0.5 * t + n * t * t:
            break
    else:
        raise AssertionError(f"{what} {key}: whole-tensor checksums {cs} disagree with the fixture's beyond what {tol:g} per element allows")
    return float(d32.max())


class GradFixture:
    def __init__(self, fix, prefix=""):
        self.names = [str(k) for k in fix[prefix + "names"]]
        self.numel, self.stride, self.offs = fix[prefix + "numel"], fix[prefix + "stride"], fix[prefix + "offs"]
        self.stats = fix[prefix + "stats"]
        self.g32 = fix[prefix + "g32"].astype(np.float64)
        self.g64 = self.g32 + fix[prefix + "d64"].astype(np.float64)
        self.index = {k: i for i, k in enumerate(self.names)}

    def entry(self, name):
        i = self.index[name]
        lo, hi = int(self.offs[i]), int(self.offs[i + 1])
        s32, s64, own, kink, _, _, sq64, sq32 = self.stats[i]
        return dict(i=i, numel=int(self.numel[i]), stride=int(self.stride[i]), g32=self.g32[lo:hi], g64=self.g64[lo:hi],
                    s32=s32, s64=s64, own=own, kink=kink, sq64=sq64, sq32=sq32)


def _sampled(t, stride):
    """Every stride-th element of the logical (OIHW) order, gathered on the device."""
    t = t.detach()
    flat = t.contiguous().reshape(-1)
    return flat[::stride].double().cpu().numpy(), float((flat.double() * flat.double()).sum().item()), float(flat.abs().max().item())


def compare_grads(got, gf, what, rule="tryon", rel=GRAD_REL, scalar_factor=None):
    """got: {name: gradient tensor}; gf: GradFixture.  Applies the r03 acceptance rule of `rule` ("tryon" =
    test_parity_bs4_gpu.compare_all_gradients, "sams" = test_sams_gpu._compare_grads) to the sampled elements, with the
    whole-tensor scales / own / kink taken from the fixture.  Returns {name: route}; one assertion lists every offender."""
    assert set(got) == set(gf.names), (what, sorted(set(got) ^ set(gf.names))[:10])
    routes, bad, rows = {}, [], []
    for name in gf.names:
        e = gf.entry(name)
        g = got[name]
        assert g.numel() == e["numel"], (what, name, g.numel(), e["numel"])
        mine, sq, gmax = _sampled(g, e["stride"])
        e32, e64 = float(np.abs(mine - e["g32"]).max()), float(np.abs(mine - e["g64"]).max())
        s32, s64, own, kink = e["s32"], e["s64"], e["own"], e["kink"]
        if rule == "tryon":
            if s64 <= 1e-6 * max(s32, 1e-12):
                ok, tag = gmax <= 10 * max(s32, GRAD_FLOOR), "zero"
            elif e32 <= rel * s32 + GRAD_FLOOR:
                ok, tag = True, "fp32"
            else:
                ok, tag = e64 <= rel * s64 + GRAD_FLOOR, "fp64"
                if not ok and g.numel() == 1:
                    ok, tag = e64 <= 30 * own, "fp64-scalar"
                if not ok and kink > 0:
                    ok, tag = min(e32, e64) <= rel * s64 + GRAD_FLOOR + 1.5 * kink, "fp64+kink"
            tol_energy = rel * max(s32, s64) + GRAD_FLOOR + 1.5 * kink + (30 * own if g.numel() == 1 else 0.0)
        else:
            big = max(s32, s64)
            if s64 <= 1e-6 * max(s32, 1e-30):
                noise = max(own, 1e-12)
                ok, tag = gmax <= 10 * noise + 1e-10 + s64, "zero"
                tol_energy = 10 * noise + 1e-10
            else:
                err = min(e32, e64)
                k_own = (10 if g.numel() == 1 else 5) * own
                if err <= 2e-3 * big:
                    ok, tag = True, "base"
                elif err <= 2e-3 * big + k_own:
                    ok, tag = True, "own"
                else:
                    ok, tag = err <= 2e-3 * big + k_own + 1.5 * kink, "kink"
                tol_energy = 2e-3 * big + k_own + 1.5 * kink
        # whole tensor: ||g||^2 against the fp64 oracle's (or the fp32 one's, whichever is nearer) within what element-wise
        # agreement at the rule's tolerance implies; catches damage between the sampled elements
        n = e["numel"]
        bound = 2 * (max(e["sq64"], e["sq32"]) * n) ** 0.5 * tol_energy + n * tol_energy ** 2
        if tag != "zero" and min(abs(sq - e["sq64"]), abs(sq - e["sq32"])) > bound:
            ok, tag = False, tag + "/energy"
        routes[name] = tag
        rows.append((name, tag, e32, s32, e64, s64, own, kink))
        if not ok:
            bad.append(rows[-1])
    tally = {}
    for r in routes.values():
        tally[r] = tally.get(r, 0) + 1
    real = [r_ for r_ in rows if r_[1] != "zero"]
    worst = max(real, key=lambda r_: min(r_[2] / max(r_[3], 1e-30), r_[4] / max(r_[5], 1e-30))) if real else None
    print(f"[{what}] {len(rows)} gradient tensors vs the committed fixture (every N-th element + energy), routes {tally}"
          + (f"; worst: {worst[0]} rel32 {worst[2] / max(worst[3], 1e-30):.1e} rel64 {worst[4] / max(worst[5], 1e-30):.1e}" if worst else ""))
    for nm, tag, e32, s32, e64, s64, own, kink in rows:
        if tag not in ("zero", "fp32", "base"):
            print(f"    {nm} [{tag}]: ours vs fp32 oracle {e32 / max(s32, 1e-30):.1e}, vs fp64 {e64 / max(s64, 1e-30):.1e}, "
                  f"fp32 oracle vs fp64 {own / max(s64, 1e-30):.1e}, kink bracket {kink / max(s64, 1e-30):.1e}")
    assert not bad, f"{what}: {len(bad)}/{len(rows)} gradient tensors out of tolerance: " + "; ".join(
        f"{nm} [{tag}] e32 {e32:.3e}/{s32:.3e} e64 {e64:.3e}/{s64:.3e} own {own:.3e} kink {kink:.3e}"
        for nm, tag, e32, s32, e64, s64, own, kink in bad[:12])
    check_routes(what, rule, routes)
    return routes


_ROUTES = None


def check_routes(what, rule, routes):
    """The committed tally pins each tensor's route: moving to a weaker one is a regression even while the rule still
    passes.  SHINEON_WRITE_ROUTES=<file> records the tallies of this run instead (merged into the file)."""
    global _ROUTES
    dump = os.environ.get("SHINEON_WRITE_ROUTES")
    if dump:
        cur = json.load(open(dump)) if os.path.exists(dump) else {}
        cur[what] = {"rule": rule, "routes": routes}
        os.makedirs(os.path.dirname(os.path.abspath(dump)), exist_ok=True)
        with open(dump, "w") as f:
            json.dump(cur, f, indent=0, sort_keys=True)
    if _ROUTES is None:
        _ROUTES = json.load(open(ROUTES_FILE)) if os.path.exists(ROUTES_FILE) else {}
    ref = _ROUTES.get(what)
    if ref is None or os.environ.get("SHINEON_ROUTES_NOCHECK"):
        return
    order = ROUTE_ORDER[rule]
    worse = [f"{k}: {ref['routes'][k]} -> {v}" for k, v in routes.items()
             if k in ref["routes"] and order.index(v.split("/")[0]) > order.index(ref["routes"][k])]
    msg = f"{what}: {len(worse)} tensors now need a weaker acceptance route than the committed tally: " + "; ".join(worse[:10])
    if ref.get("strict", True):
        # every layer shape of these cases has a COMMITTED igemm plan (plans/gfx950.txt): same kernels, same summation order,
        # same bits on every box - a route change is a code change
        assert not worse, msg
    elif worse:
        # SAMS cases: some of their layer shapes are measured per process (no committed plan), so the split-K order - and with
        # it a tensor sitting on a route boundary - may differ between boxes: reported, not failed
        import warnings

        warnings.warn(msg)


def check_scalar(ours, r32, r64, tol_abs, tol_rel, what):
    ours, r32 = float(ours), float(r32)
    cands = [r32] + ([float(r64)] if r64 is not None else [])
    err = min(abs(ours - c) for c in cands)
    assert err <= tol_abs + tol_rel * abs(cands[-1]), (what, ours, cands)
