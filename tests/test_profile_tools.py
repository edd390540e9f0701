"""tools/pmc_traffic.py turns the rocprofv3 --pmc passes of tools/profile_round.sh into the per-STEP table bench.py reads back as
`roofline.traffic`.  The passes run `bench.py --steps 1 --warmup 0`, which goes through the hot path 1 + SERIAL_TABLE_STEPS times; the
divisor once stayed at 2 when the kernel table became a median of three steps (round 4: traffic reported twice too large) — the tool now
takes the constant from bench.py and checks it against the launches of k_parts."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pass(dirname, counter, rows):
    os.makedirs(dirname, exist_ok=True)
    with open(os.path.join(dirname, "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["Counter_Name", "Kernel_Name", "Counter_Value"])
        w.writeheader()
        for kernel, value, launches in rows:
            for _ in range(launches):
                w.writerow({"Counter_Name": counter, "Kernel_Name": kernel, "Counter_Value": value / launches})


def test_pmc_traffic_divides_by_the_runs_of_the_command(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    runs = 1 + bench.SERIAL_TABLE_STEPS
    line = {"config": {"genome_bp": 3_000_000_000, "reads_per_gpu": 10_000_000, "read_len": 150, "k": 4}}
    (tmp_path / "line.json").write_text(json.dumps(line) + "\n")
    sub = 3  # (10^7 reads: three sub-batches, one k_parts launch each per step)
    _pass(str(tmp_path / "f"), "FETCH_SIZE", [("void cmb::k_parts<2, false, 8>(int)", 8000.0 * runs, sub * runs), ("void cmb::k_bfs_pass<cmb::GeoN32>(x)", 400.0 * runs, 432 * runs),
                                              ("at::native::something(float)", 1e9, 5)])
    _pass(str(tmp_path / "w"), "WRITE_SIZE", [("void cmb::k_parts<2, false, 8>(int)", 100.0 * runs, sub * runs), ("void cmb::k_bfs_pass<cmb::GeoN32>(x)", 60.0 * runs, 432 * runs)])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "line.json"), str(tmp_path / "f"), str(tmp_path / "w")],
                         capture_output=True, text=True, check=True)
    d = json.loads(out.stdout)
    assert d["steps_run"] == runs and d["kernel_src_sha"] == bench.source_digest()
    assert d["kernels"]["k_bfs_pass"] == {"FETCH_SIZE_KiB": 400.0, "dispatches_per_step": 432.0, "WRITE_SIZE_KiB": 60.0}
    assert d["kernels"]["k_parts"]["dispatches_per_step"] == sub and "something" not in "".join(d["kernels"])
    # the same passes read with the wrong number of runs: refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "line.json"), str(tmp_path / "f"), str(tmp_path / "w"), "2"],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "wrong number of steps" in bad.stderr
