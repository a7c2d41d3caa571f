"""Host-side checks of the committed full-size fixtures (tests/golden/full/) and of the acceptance machinery in tests/gradfix.py:
the fixtures load, belong to the inputs the GPU tests rebuild (digest), carry every case the GPU tests ask for, and the rules
accept the oracle's own numbers while rejecting damaged ones - on the sampled elements AND between them (energy bound)."""
import json
import os

import numpy as np
import pytest
import torch

import fullsize_cases as fc
import gradfix as gf

CASES = ["warp_bs2", "chain_bs4", "chain_bs8", "c5", "sams_base", "sams_attn_gelu", "sams_progressive", "sams_full_generator_bs4",
         "sams_full_three_steps"]


@pytest.mark.parametrize("name", CASES)
def test_fixture_loads_and_is_complete(name):
    fix = gf.load(name)
    prefixes = {"warp_bs2": ["grad:"], "chain_bs4": ["warp:grad:", "tryon:grad:"], "chain_bs8": ["warp:grad:", "tryon:grad:"],
                "c5": ["grad:"], "sams_full_generator_bs4": ["grad:"]}.get(name, ["grad0:", "grad1:", "grad2:"])
    for pfx in prefixes:
        g = gf.GradFixture(fix, pfx)
        assert len(g.names) > 0 and g.offs[-1] == g.g32.size == g.g64.size
        for i, k in enumerate(g.names):
            e = g.entry(k)
            n = e["g32"].size
            assert n == len(range(0, e["numel"], e["stride"])) and (e["stride"] == 1 or n <= gf.SAMPLES), k
            assert e["s32"] >= 0 and e["own"] >= 0 and e["kink"] >= 0
            assert np.abs(e["g32"]).max() <= e["s32"] * (1 + 1e-6) + 1e-30, k      # samples lie inside the whole-tensor maximum
    assert any(k.startswith("digest:") or ":digest:" in k for k in fix.files)
    # the fp32 leg of every full-size fixture is the REFERENCE's own evaluation (imported in the build container through
    # tests/golden/make_golden.py's shim), not the oracle's fp32 pass: try-on / warp cases since round 5, SAMS since round 6
    src = [str(fix[k]) for k in fix.files if k == "fp32_source" or k.endswith(":fp32_source")]
    assert src and all(s == "reference" for s in src), (name, src)


def test_fixture_digest_matches_the_inputs_the_gpu_tests_rebuild():
    fix = gf.load("chain_bs4")
    gf.check_digest(fix, "warp:digest:batch", fc.smooth_batch(4))
    _, wsd = fc.build_warp()
    gf.check_digest(fix, "warp:digest:weights", wsd)
    batch = dict(fc.smooth_batch(4))
    batch["cloth"] = torch.from_numpy(fix["warp:handoff_f16"]).float()
    gf.check_digest(fix, "tryon:digest:batch", batch)
    with pytest.raises(AssertionError, match="procedural generator changed"):
        gf.check_digest(fix, "warp:digest:batch", fc.smooth_batch(4, seed=421))


def test_route_tally_is_committed_for_every_gradient_case():
    routes = json.load(open(gf.ROUTES_FILE))
    for what in ("chained/warp bs=4 (graph replay)", "chained/try-on bs=4 (graph replay)", "chained/warp bs=8 (graph replay)",
                 "chained/try-on bs=8 (graph replay)", "C5 n_frames=5 flow_warp ngf=167", "WarpModel bs=4", "sams full size step 0",
                 "generator bs=4 full size", "base step 0", "C5 bs=2 (graph replay)"):
        assert what in routes and routes[what]["rule"] in gf.ROUTE_ORDER, what
        assert set(routes[what]["routes"].values()) <= set(gf.ROUTE_ORDER[routes[what]["rule"]]), what
    # round 5: every case - the SAMS ones included - runs on COMMITTED igemm plans (tools/gpu_make_plans.sh collects the
    # shapes of the tests themselves), so every tally is pinned strictly: a route change fails on every box
    assert all(v.get("strict", True) for v in routes.values()), [k for k, v in routes.items() if not v.get("strict", True)]


def _synthetic_fixture():
    g = torch.Generator().manual_seed(3)
    g64 = {"w": torch.randn(40, 30, 3, 3, generator=g, dtype=torch.float64), "b": torch.randn(40, generator=g, dtype=torch.float64) * 1e-3,
           "zero": torch.randn(16, generator=g, dtype=torch.float64) * 1e-16}
    g32 = {k: (v + torch.randn(v.shape, generator=g, dtype=torch.float64) * 1e-6 * v.abs().max()).float() for k, v in g64.items()}
    g32["zero"] = (torch.randn(16, generator=g, dtype=torch.float64) * 1e-9).float()   # round-off noise where the exact value is 0
    out = {}
    gf.pack_grads(out, "g:", g32, g64, {"w": 0.0, "b": 0.0, "zero": 0.0})
    return out, g32, g64


@pytest.mark.parametrize("rule", ["tryon", "sams"])
def test_rules_accept_the_oracle_and_reject_damage(rule, monkeypatch):
    monkeypatch.setenv("SHINEON_ROUTES_NOCHECK", "1")
    out, g32, g64 = _synthetic_fixture()
    gfx = gf.GradFixture(out, "g:")
    assert gfx.entry("w")["stride"] > 1            # 10 800 elements: sampled
    routes = gf.compare_grads({k: v.clone() for k, v in g32.items()}, gfx, "self", rule=rule)
    assert routes["w"] in ("fp32", "base") and routes["zero"] == "zero"
    bad = {k: v.clone() for k, v in g32.items()}
    bad["w"].view(-1)[0] += 0.05 * g32["w"].abs().max()      # a SAMPLED element (index 0 is always on the lattice)
    with pytest.raises(AssertionError, match="out of tolerance"):
        gf.compare_grads(bad, gfx, "damaged sample", rule=rule)
    st = gfx.entry("w")["stride"]
    bad = {k: v.clone() for k, v in g32.items()}
    flat = bad["w"].view(-1)
    off = [i for i in range(flat.numel()) if i % st][:2000]       # 2000 elements BETWEEN the samples replaced by garbage
    flat[off] = 3.0 * g32["w"].abs().max()
    with pytest.raises(AssertionError, match="energy"):
        gf.compare_grads(bad, gfx, "damaged between samples", rule=rule)
    noisy = {k: v.clone() for k, v in g32.items()}
    noisy["zero"] = torch.full((16,), 1e-3)                        # an analytically-zero gradient must stay at noise level
    with pytest.raises(AssertionError, match="out of tolerance"):
        gf.compare_grads(noisy, gfx, "zero rule", rule=rule)


def test_output_rule_modes():
    t64 = torch.randn(2, 3, 24, 18, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    t32 = (t64 + 6e-5 * torch.sign(torch.randn(t64.shape, generator=torch.Generator().manual_seed(6)))).float()
    out = {}
    gf.pack_output(out, "y", t32, t64)

    class F(dict):
        files = property(lambda self: list(self.keys()))

    fix = F(out)
    gf.check_output(fix, "y", t32, 1e-4, "self", mode="fp32")
    exactish = t64.float()                                            # 6e-5 from the fp32 "reference", 0 from exact
    gf.check_output(fix, "y", exactish, 1e-4, "exact", mode="either")
    far = (t64 + 1.3e-4).float()                                       # 1.3e-4 from exact: outside fp32 rule, inside either (1e-4 + 6e-5)
    with pytest.raises(AssertionError):
        gf.check_output(fix, "y", far, 1e-4, "far", mode="fp32")
    with pytest.raises(AssertionError, match="twice"):
        gf.check_output(fix, "y", far, 1e-4, "far", mode="either")     # ... but more than twice as far from exact as the reference
    hole = t32.clone()
    for r_ in (1, 2, 4, 5, 7, 8):
        hole[0, 0, r_, [1, 2, 4, 5, 7, 8]] = 7.0                       # garbage OFF the 1/9 lattice: the checksums catch it
    with pytest.raises(AssertionError, match="checksums"):
        gf.check_output(fix, "y", hole, 1e-4, "hole", mode="fp32")
