import csv, collections, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
t_end = int(rows[-1]["End_Timestamp"])
win = [r for r in rows if int(r["Start_Timestamp"]) > t_end - int(ms * 1e6)]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
print("last %.0f ms: kernels %d busy %.2f ms" % (ms, len(win), busy / 1e6))
acc = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = r["Kernel_Name"]
    n = re.sub(r"at::native::|\(anonymous namespace\)::|std::array<char\*, \d+ul>|unsigned int|TensorIteratorBase&, ", "", n)
    acc[n[:170]][0] += 1; acc[n[:170]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:70]:
    print("%5d %8.2f ms  %s" % (v[0], v[1] / 1e6, k))
print("longest kernels in the window:")
for r in sorted(win, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:25]:
    n = re.sub(r"at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:90]
    print("  %8.1f us  grid %sx%sx%s wg %s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], n))
