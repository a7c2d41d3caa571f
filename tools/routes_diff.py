"""Tally diff of two route files (tests/golden/full/routes.json format): per case, how many gradient tensors sit on each
acceptance route, and every tensor whose route changed (weaker / stronger in the rule's own order, tests/gradfix.ROUTE_ORDER).
    python tools/routes_diff.py tests/golden/full/routes.json gpurun_out/routes_new.json"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gradfix as gf  # noqa: E402

old, new = (json.load(open(p)) for p in sys.argv[1:3])
weaker = stronger = 0
for case in sorted(set(old) | set(new)):
    if case not in old or case not in new:
        print(f"{case}: only in {'new' if case in new else 'old'}")
        continue
    order = gf.ROUTE_ORDER[new[case]["rule"]]
    ro, rn = old[case]["routes"], new[case]["routes"]
    tally = lambda r: dict(collections.Counter(v.split("/")[0] for v in r.values()))  # noqa: E731
    moved = []
    for k in sorted(set(ro) & set(rn)):
        a, b = order.index(ro[k].split("/")[0]), order.index(rn[k].split("/")[0])
        if a != b:
            moved.append(f"    {'WEAKER  ' if b > a else 'stronger'} {k}: {ro[k]} -> {rn[k]}")
            weaker += b > a
            stronger += b < a
    print(f"{case}: {tally(ro)} -> {tally(rn)}" + (f"  ({len(set(rn) - set(ro))} new, {len(set(ro) - set(rn))} gone)" if set(ro) ^ set(rn) else ""))
    print("\n".join(moved)) if moved else None
print(f"total: {weaker} tensors moved to a weaker route, {stronger} to a stronger one")
