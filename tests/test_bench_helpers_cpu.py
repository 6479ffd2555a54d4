"""bench.py's host-side helpers (no GPU): the module imports, the CPU partition of the ranks is disjoint and socket-contiguous, the committed PMC summaries the
line quotes are readable."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_imports_and_reads_its_committed_summaries():
    sys.path.insert(0, ROOT)
    import bench
    assert os.environ.get("GPU_MAX_HW_QUEUES")            # one hardware queue per driver handle, set before the HIP runtime starts
    cores = bench.physical_cores()
    assert cores and all(len(g) >= 1 for g in cores)
    t = bench._lba_traffic()
    assert isinstance(t, int) and t > 1 << 20               # profiles/r03_pmc_lba_traffic.json: tens of MB per launch
    m = bench._mfma_counters()
    assert m is None or 0 < m["mfma_util"] < 1


def test_rank_cpu_sets_are_disjoint():
    """pin_rank_cpus in child processes (it changes the affinity of the caller): the sets of the ranks of one node do not overlap and cover whole physical cores."""
    code = ("import sys, os, json; sys.path.insert(0, %r); import bench; n = bench.pin_rank_cpus(int(sys.argv[1]), int(sys.argv[2]), 0); "
            "print(json.dumps(sorted(os.sched_getaffinity(0))))" % ROOT)
    world = 2 if len(os.sched_getaffinity(0)) >= 2 else 1
    sets = []
    for r in range(world):
        out = subprocess.run([sys.executable, "-c", code, str(r), str(world)], capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1]
        sets.append(set(json.loads(out)))
    assert all(sets)
    if world == 2 and len({tuple(g) for g in __import__("bench").physical_cores()}) >= 2:
        assert not (sets[0] & sets[1])


def test_eight_ranks_on_a_two_socket_node_get_whole_cores_of_one_socket_each():
    """configs[4] (8 GPUs, one rank each) on the topology of the MI355X hosts — 2 sockets x 64 cores x 2 SMT threads = 256 CPUs, siblings numbered c and c + 128:
    every rank gets 16 physical cores = 32 CPUs (>= the 16 host threads a rank runs), disjoint from the others, SMT siblings together, all on one socket."""
    sys.path.insert(0, ROOT)
    import bench
    cores = [[c, c + 128] for c in range(128)]                  # (package, core) order: cores 0-63 on socket 0, 64-127 on socket 1
    sets = [bench.rank_cpu_set(cores, r, 8) for r in range(8)]
    assert all(len(s) == 32 for s in sets)
    assert len(set().union(*map(set, sets))) == 256             # disjoint and complete
    for r, s in enumerate(sets):
        phys = sorted(c for c in s if c < 128)
        assert sorted(c - 128 for c in s if c >= 128) == phys   # both threads of every core
        assert phys == list(range(16 * r, 16 * r + 16))         # contiguous
        assert len({c // 64 for c in phys}) == 1                # one socket
    assert len(bench.rank_cpu_set(cores, 3, 8, per_rank_cores=8)) == 16
