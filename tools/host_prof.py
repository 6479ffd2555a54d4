"""Where the host cores go in the batch-of-sequences driver: PC samples (tools/pcsample) of the headline workload's timed steps, attributed to functions.
  on the GPU box : python tools/host_prof.py run [S=4096] [G=8] [steps=10]      -> gpurun_out/host_prof.samples
  afterwards     : python tools/host_prof.py report gpurun_out/host_prof.samples [top=60]"""
import bisect, collections, ctypes, os, subprocess, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")   # as bench.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(S, G, steps, preroll=200):
    from object_slam_amd import seqbench, slam
    wl = seqbench.rgbd_workload(speed=1.0, n_base=8, stagger=24)                  # bench.py's headline: steady state after `preroll` steps
    warm = 4
    seqs = seqbench.base_sequences(wl, 0, S, preroll + warm + steps, workers=16)   # forked workers, before the GPU is touched
    so = os.path.join(ROOT, "tools", "pcsample", "libpcsample.so")
    if not os.path.exists(so):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tools", "pcsample", "pcsample.c"), "-lrt"])
    pcs = ctypes.CDLL(so)
    share = min(16, os.cpu_count() or 1)
    summ, rec, systems, _ = seqbench.run_rank(wl, lambda cfg: slam.System(cfg), 0, 1, S, G, steps, warm, True, 0, host_threads=max(1, share // G),
                                              sequences=seqs, after_warmup=lambda systems: pcs.pcs_start(997), preroll=preroll,
                                              local_mapping=slam.LM_SYNC if os.environ.get("HOST_PROF_SYNC") else slam.LM_DEFERRED,   # bench.py's default schedule
                                              progress=lambda m: print(m, flush=True))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    n = pcs.pcs_stop(os.path.join(ROOT, "gpurun_out", "host_prof.samples").encode())
    print("frames/s", round(summ["frames_per_s"], 1), "samples", n)


def exec_delta(lib):
    """p_vaddr - p_offset of the executable LOAD segment: /proc/self/maps gives file offsets, the symbol and line tables virtual addresses."""
    try:
        out = subprocess.run(["readelf", "-lW", lib], capture_output=True, text=True).stdout
        for l in out.splitlines():
            f = l.split()
            if len(f) >= 7 and f[0] == "LOAD" and "E" in "".join(f[6:-1]):
                return int(f[2], 16) - int(f[1], 16)
    except Exception:
        pass
    return 0


def report(path, top):
    maps, samples = [], []
    for line in open(path):
        if line[0] == "M":
            f = line[2:].split()
            lo, hi = (int(x, 16) for x in f[0].split("-"))
            maps.append((lo, hi, int(f[2], 16), f[5] if len(f) > 5 else "[anon]"))
        else:
            samples.append(int(line[2:], 16))
    maps.sort()
    starts = [m[0] for m in maps]
    per_lib = collections.Counter()
    per_fn = collections.Counter()
    syms = {}

    deltas = {}

    def delta(lib):
        if lib not in deltas:
            cand = lib if os.path.exists(lib) else os.path.join(ROOT, "object_slam_amd", os.path.basename(lib))
            deltas[lib] = exec_delta(cand) if os.path.exists(cand) else 0
        return deltas[lib]

    def table(lib):
        if lib in syms:
            return syms[lib]
        cand = lib if os.path.exists(lib) else os.path.join(ROOT, "object_slam_amd", os.path.basename(lib))
        tab = []
        if os.path.exists(cand):
            for flags in (["-C", "--defined-only"], ["-C", "-D", "--defined-only"]):
                try:
                    out = subprocess.run(["nm"] + flags + [cand], capture_output=True, text=True).stdout
                except Exception:
                    out = ""
                for l in out.splitlines():
                    p = l.split(" ", 2)
                    if len(p) == 3 and p[1] in "tTwW":
                        tab.append((int(p[0], 16), p[2]))
        tab.sort()
        syms[lib] = ([a for a, _ in tab], [n for _, n in tab])
        return syms[lib]

    for pc in samples:
        i = bisect.bisect_right(starts, pc) - 1
        if i < 0 or pc >= maps[i][1]:
            per_lib["?"] += 1
            continue
        lo, hi, off, lib = maps[i]
        base = os.path.basename(lib)
        per_lib[base] += 1
        if "oslam" in base or "pcsample" in base:
            addrs, names = table(lib)
            a = pc - lo + off + delta(lib)
            j = bisect.bisect_right(addrs, a) - 1
            per_fn[names[j][:150] if j >= 0 else "?"] += 1
    # line-level attribution when the library was built with line tables (OSLAM_EXTRA_FLAGS=-gline-tables-only python -m object_slam_amd.build)
    per_line, per_outer = collections.Counter(), collections.Counter()
    ours = collections.defaultdict(list)
    for pc in samples:
        i = bisect.bisect_right(starts, pc) - 1
        if i >= 0 and pc < maps[i][1] and "oslam" in os.path.basename(maps[i][3]):
            ours[maps[i][3]].append(pc - maps[i][0] + maps[i][2] + delta(maps[i][3]))
    for lib, addrs in ours.items():
        cand = lib if os.path.exists(lib) else os.path.join(ROOT, "object_slam_amd", os.path.basename(lib))
        if not os.path.exists(cand):
            continue
        sym = os.environ.get("LLVM_SYMBOLIZER", "/opt/rocm/lib/llvm/bin/llvm-symbolizer")
        out = subprocess.run([sym, "--obj=" + cand, "--output-style=JSON"], input="\n".join(hex(a) for a in addrs), capture_output=True, text=True).stdout
        import json
        for l in out.splitlines():
            try:
                fr = json.loads(l).get("Symbol", [])
            except Exception:
                continue
            if not fr:
                continue
            inner, outer = fr[0], fr[-1]
            per_line["%s:%s" % (os.path.basename(inner.get("FileName", "?")), inner.get("Line", 0))] += 1
            per_outer["%s:%s" % (os.path.basename(outer.get("FileName", "?")), outer.get("Line", 0))] += 1
    n = len(samples)
    print("samples", n)
    for k, v in per_lib.most_common(12):
        print("  %6.2f %%  %s" % (100.0 * v / n, k))
    print("functions of the library:")
    for k, v in per_fn.most_common(top):
        print("  %6.2f %%  %s" % (100.0 * v / n, k))
    if per_line and not (len(per_line) == 1 and "??" in next(iter(per_line))):
        print("source lines (innermost inlined frame):")
        for k, v in per_line.most_common(top):
            print("  %6.2f %%  %s" % (100.0 * v / n, k))
        print("source lines (outermost frame = line of the non-inlined function the sample lies in):")
        for k, v in per_outer.most_common(top * 2):
            print("  %6.2f %%  %s" % (100.0 * v / n, k))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        a = [int(x) for x in sys.argv[2:]]
        run(*(a + [4096, 8, 10][len(a):]))
    else:
        report(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 60)
