"""Static instruction statistics of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S).
usage: python tools/isa_stats.py file.s kernel_name_substring"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + sys.argv[2] + r"\w*:", l)][0]
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
ins = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
valu = sum(v for k, v in c.items() if k.startswith("v_")); salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop")))
print("instructions %d: valu %d salu %d branch %d lds %d vmem %d waitcnt %d" % (len(ins), valu, salu, sum(v for k, v in c.items() if k.startswith(("s_cbranch", "s_branch"))),
      sum(v for k, v in c.items() if k.startswith("ds_")), sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "scratch_", "flat_"))), c["s_waitcnt"]))
print(c.most_common(12))
for l in lines[end:end + 80]:
    if re.search(r"; (NumVgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs)", l): print(l.strip())
