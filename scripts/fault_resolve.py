"""Reads a log that holds the ROCm runtime's "Memory access fault by GPU ... on address X" line and a SIMKA_FAULT_TRACE dump
(simka_amd/csrc/simka_trace.h) and says what X is: the range of the library that holds it (or the nearest ones, with the distance), whether
memory was mapped there at the time of the fault, and which kernels were launched last (per stream).
usage: fault_resolve.py LOG [DUMP]      (DUMP: the simka_fault_trace.<pid>.txt file, when stderr was not captured in LOG)"""
import re
import sys


def parse(text):
    faults = [int(m.group(1), 16) for m in re.finditer(r"Memory access fault by GPU .* on address (0x[0-9a-fA-F]+)", text)]
    ranges = []
    for m in re.finditer(r"\[simka-trace\] range (0x[0-9a-f]+) - (0x[0-9a-f]+)\s+(\d+) B (\w+)\s+added@([\d.]+) (live|freed@[\d.]+)\s+(.*?) \((.*?):(\d+)\)", text):
        ranges.append(dict(lo=int(m.group(1), 16), hi=int(m.group(2), 16), kind=m.group(4), added=float(m.group(5)),
                           freed=None if m.group(6) == "live" else float(m.group(6)[6:]), name=m.group(7), file=m.group(8), line=int(m.group(9))))
    launches = []
    for m in re.finditer(r"\[simka-trace\] launch #(\d+) t=([\d.]+) \(([\d.]+) ms ago\) stream (0x[0-9a-f]+) grid (\d+) x (\d+) block (\d+) lds (\d+) ctx (\d+) sample (\d+) "
                         r"arena (\d+) / (\d+) / (\d+)  (.*)", text):
        launches.append(dict(seq=int(m.group(1)), t=float(m.group(2)), ago_ms=float(m.group(3)), stream=m.group(4), grid=(int(m.group(5)), int(m.group(6))),
                             block=int(m.group(7)), lds=int(m.group(8)), ctx=int(m.group(9)), sample=int(m.group(10)),
                             arena=(int(m.group(11)), int(m.group(12)), int(m.group(13))), kernel=m.group(14).strip()))
    m = re.search(r"\[simka-trace\] ==== signal (\d+) at t = ([\d.]+) s", text)
    return faults, ranges, launches, (float(m.group(2)) if m else None)


def resolve(addr, ranges, t_fault):
    """-> list of text lines"""
    out = []
    inside = [r for r in ranges if r["lo"] <= addr < r["hi"]]
    for r in sorted(inside, key=lambda r: r["hi"] - r["lo"]):
        state = "live" if r["freed"] is None else "FREED %.3f ms before the fault" % ((t_fault - r["freed"]) * 1e3 if t_fault else 0.0)
        age = " (added %.3f ms before the fault)" % ((t_fault - r["added"]) * 1e3) if t_fault else ""
        out.append("  inside %-8s %s [%s:%d]  offset %d of %d bytes, %s%s" % (r["kind"], r["name"], r["file"], r["line"], addr - r["lo"], r["hi"] - r["lo"], state, age))
    reserved = [r for r in inside if r["kind"] == "reserved"]
    mapped = [r for r in inside if r["kind"] == "chunk" and r["freed"] is None]
    if reserved and not mapped:
        out.append("  => a reserved virtual range with NO memory mapped at this address: the arena was touched beyond (or before) its mapped chunks")
    if not inside:
        below = [r for r in ranges if r["hi"] <= addr]
        above = [r for r in ranges if r["lo"] > addr]
        if below:
            r = max(below, key=lambda r: r["hi"])
            out.append("  not inside any range of the library; %d bytes past the end of %s %s [%s:%d] (%s)" % (addr - r["hi"], r["kind"], r["name"], r["file"], r["line"],
                                                                                                         "live" if r["freed"] is None else "freed"))
        if above:
            r = min(above, key=lambda r: r["lo"])
            out.append("  %d bytes before the start of %s %s [%s:%d]" % (r["lo"] - addr, r["kind"], r["name"], r["file"], r["line"]))
        if not below and not above:
            out.append("  no ranges in the dump")
    return out


def main():
    text = open(sys.argv[1], errors="replace").read()
    if len(sys.argv) > 2:
        text += "\n" + open(sys.argv[2], errors="replace").read()
    faults, ranges, launches, t_fault = parse(text)
    if not ranges and not launches:
        print("no SIMKA_FAULT_TRACE dump in the input (run with SIMKA_FAULT_TRACE=1)")
        return 2
    print("%d ranges, %d launches in the dump" % (len(ranges), len(launches)))
    for a in faults or []:
        print("fault address 0x%x:" % a)
        for ln in resolve(a, ranges, t_fault):
            print(ln)
    if not faults:
        print("no 'Memory access fault' line in the input")
    last = {}
    for l in launches:
        last.setdefault(l["stream"], []).append(l)
    for st, ls in last.items():
        print("stream %s: last launches" % st)
        for l in ls[-4:]:
            print("  #%d %8.3f ms before the dump  %s  grid %s block %d ctx %d sample %s  arena mapped %d / bound %d / cap %d" % (
                l["seq"], l["ago_ms"], l["kernel"], "x".join(map(str, l["grid"])), l["block"], l["ctx"], "merge" if l["sample"] == 0xffffffff else l["sample"], *l["arena"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
