"""Instruction mix of the main (MFMA) loop of every so_igemm_kernel instantiation in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/igemm2.s igemm2.hip
    python tools/isa_loop_mix.py /tmp/igemm2.s [name-filter]
"""
import re
import sys


def analyze(s, name):
    i = s.index(name + ":")
    j = s.index("s_endpgm", i)
    lines = s[i:j].split("\n")
    mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
    labels = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = k
    best = None
    for k, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            a, b = labels[m.group(1)], k
            n = sum(1 for x in mf if a < x < b)
            if n >= 8 and (best is None or n > best[2]):
                best = (a, b, n)
    if best is None:
        return None
    a, b, _ = best
    body = [l.strip() for l in lines[a:b] if l.strip() and not l.strip().startswith((".", ";"))]
    cnt = {}
    for l in body:
        op = l.split()[0]
        if op.startswith("v_mfma"):
            key = "mfma"
        elif op.startswith("v_"):
            key = "valu"
        elif op.startswith("s_waitcnt"):
            key = "waitcnt"
        elif op.startswith("s_"):
            key = "salu"
        elif op.startswith("buffer_"):
            key = "buffer_load"
        else:
            key = op
        cnt[key] = cnt.get(key, 0) + 1
    return cnt


if __name__ == "__main__":
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for name in sorted(set(re.findall(r"^(_Z15so_igemm_kernel\w+):", s, re.M))):
        if flt in name:
            c = analyze(s, name)
            if c:
                print(name[19:-9], "valu/mfma=%.1f" % (c.get("valu", 0) / max(1, c.get("mfma", 1))), c)
