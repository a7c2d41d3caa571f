"""Prints the order of memory / MFMA / wait instructions in the main loop of one so_igemm_kernel instantiation.
M mfma, r ds_read, w ds_write, G buffer_load, B barrier, . VALU, , SALU, |..| s_waitcnt.
    python tools/isa_loop_order.py /tmp/igemm2.s ILi0ELb0ELb0ELi64ELi64ELi4E"""
import re
import sys

s = open(sys.argv[1]).read()
name = next(n for n in re.findall(r"^(_Z15so_igemm_kernel\w+):", s, re.M) if sys.argv[2] in n)
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
a, b, _ = best
seq = []
for l in lines[a:b]:
    t = l.strip()
    if not t or t.startswith((".", ";")):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"):
        seq.append("M")
    elif op.startswith("ds_read"):
        seq.append("r")
    elif op.startswith("ds_write"):
        seq.append("w")
    elif op.startswith("buffer_load"):
        seq.append("G")
    elif op.startswith("s_waitcnt"):
        seq.append("|" + t.split(None, 1)[1].replace(" ", "") + "|")
    elif op.startswith("s_barrier"):
        seq.append("B")
    elif op.startswith("v_"):
        seq.append(".")
    elif op.startswith("s_"):
        seq.append(",")
print(name)
print("".join(seq))
