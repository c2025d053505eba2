"""The digest scripts behind profiles/ (tools/pmc_requests_summary.py, tools/extract_kernel_rates.py) on small hand-made
rocprofv3 CSVs: what `bench.py`'s roofline_sample and the extraction's kernel-duration rates are computed from."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(path, header, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(",".join(header) + "\n")
        for r in rows:
            f.write(",".join('"%s"' % x for x in r) + "\n")


def test_pmc_requests_summary_counts_requests_per_batch_and_kernel(tmp_path):
    hdr = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    rows, d = [], 0
    for b in range(4):  # four batches: two sampler launches, one split, one gather each
        for name, rd, wr in (("void fgnn::(anonymous namespace)::khop_sample_kernel<64, 256, true, 16>(x)", 100, 50),
                             ("void fgnn::(anonymous namespace)::khop_sample_kernel<64, 256, true, 32>(x)", 300, 250),
                             ("fgnn::(anonymous namespace)::cache_split_fused_kernel(x)", 400, 80),
                             ("void fgnn::(anonymous namespace)::gather_rows16_kernel<4, 32, true, true>(x)", 1000, 2000)):
            d += 1
            rows += [(d, name, "TCC_EA0_RDREQ_sum", rd), (d, name, "TCC_EA0_WRREQ_sum", wr)]
    d += 1  # one set-up kernel (not part of a batch) and the probe's launches: 1 + 24 launches of 4 M reads each
    rows += [(d, "fgnn::(anonymous namespace)::tree_fill_kernel(x)", "TCC_EA0_RDREQ_sum", 12345),
             (d, "fgnn::(anonymous namespace)::tree_fill_kernel(x)", "TCC_EA0_WRREQ_sum", 1)]
    for _ in range(25):
        d += 1
        rows += [(d, "random_read_probe_kernel(unsigned int const*, ...)", "TCC_EA0_RDREQ_sum", 4_000_000),
                 (d, "random_read_probe_kernel(unsigned int const*, ...)", "TCC_EA0_WRREQ_sum", 0)]
    _rows(str(tmp_path / "wl_req" / "x" / "1_counter_collection.csv"), hdr, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_requests_summary.py"), str(tmp_path), "wl"],
                         capture_output=True, text=True, check=True).stdout
    r = json.loads(out)["workloads"]["wl"]
    assert r["batches"] == 4 and r["probe_requests_per_read"] == 1.0
    assert r["sampler_side_per_batch"] == {"read": 800.0, "write": 380.0}  # the gather and the set-up kernel stay out
    assert r["kernels"]["khop_sample_kernel"] == {"launches_per_batch": 2.0, "read_per_batch": 400.0, "write_per_batch": 300.0}
    assert "tree_fill_kernel" not in r["kernels"] and r["kernels"]["gather_rows16_kernel"]["write_per_batch"] == 2000.0


def test_extract_kernel_rates_uses_the_union_of_overlapping_launches(tmp_path):
    hdr = ["Kernel_Name", "Start_Timestamp", "End_Timestamp"]
    # 16 launches of 400 us each, a new one every 100 us: four overlap at any time, the link is busy all along
    rows = [("void fgnn::(anonymous namespace)::extract_fused_kernel<4, 4, 32>(fgnn::FusedArgs)", 100_000 * k, 100_000 * k + 400_000)
            for k in range(16)]
    rows.append(("other_kernel", 0, 10))
    _rows(str(tmp_path / "p" / "7_kernel_trace.csv"), hdr, rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "extract_kernel_rates.py"), str(tmp_path), "5000000",
                          "100000000"], capture_output=True, text=True, check=True).stdout
    line = [ln for ln in out.splitlines() if "kernel_trace" in ln][0]
    cells = [c.strip() for c in line.strip("|").split("|")]
    # the steady part = the last 12 launches: 400 us each, busy (11 x 100 + 400) / 12 = 125 us per launch -> 40 GB/s
    assert cells[1] == "12" and cells[2] == "400.0" and cells[3] == "125.0" and cells[4] == "40.0"
