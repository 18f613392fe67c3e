"""rocprofv3 --pmc passes of an arbitrary command of this repo; mean per-launch counters of the kernels matching a substring.
usage: pmc_run.py <kernel substring> <skip first N launches> <out.json> -- <command ...>
Each pass has a deadline and one retry (the tool hangs now and then on this pool); counters of a failed pass are simply absent."""
import csv, glob, json, os, shutil, subprocess, sys, tempfile, time

PASSES = [["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"],
          ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT"],
          ["TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"]]
pat, skip, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cmd = sys.argv[sys.argv.index("--") + 1:]
exe = shutil.which("rocprofv3")
base = tempfile.mkdtemp(prefix="pmcrun_", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
out, info, failed = {}, {}, []
for i, grp in enumerate(PASSES):
    d = os.path.join(base, f"p{i}")
    ok = False
    for attempt in range(2):
        shutil.rmtree(d, ignore_errors=True)
        pr = subprocess.Popen([exe, "--pmc"] + grp + ["--kernel-include-regex", pat, "-f", "csv", "-d", d, "-o", "b", "--"] + cmd, cwd=os.getcwd(), env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
        try:
            pr.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(pr.pid, 9)
            except OSError:
                pass
            pr.wait()
            continue
        if pr.returncode == 0:
            ok = True
            break
    if not ok:
        failed.append("+".join(grp))
        continue
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat not in row["Kernel_Name"]:
                continue
            per.setdefault(row["Counter_Name"], []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            info = {"kernel": row["Kernel_Name"], "vgpr": int(row.get("VGPR_Count", 0) or 0), "sgpr": int(row.get("SGPR_Count", 0) or 0),
                    "scratch": int(row.get("Scratch_Size", 0) or 0), "lds": int(row.get("LDS_Block_Size", 0) or 0)}
    for c, v in per.items():
        v.sort()
        vals = [x for _, x in v][skip:]
        if vals:
            out[c] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
shutil.rmtree(base, ignore_errors=True)
# the kernel's registers / spills / scratch from the code object's own notes (rocprofv3's VGPR_Count is an allocation figure, not the descriptor's)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    import elf_resources
    elf = [k for k in elf_resources.kernels() if k["kernel"] == info.get("kernel")]
    info["elf"] = elf[0] if elf else None
    if elf:
        info["elf"].pop("kernel", None)
except Exception as e:  # the profile is still worth having without it
    info["elf"] = {"error": repr(e)}
import hashlib
info["library_sha16"] = hashlib.sha256(open(elf_resources.DEFAULT, "rb").read()).hexdigest()[:16]  # the library the counters were read from
info["commit"] = os.environ.get("SF_COMMIT")  # set by the gpurun wrapper (`SF_COMMIT=$(git rev-parse --short HEAD)`): .git does not travel
res = {"command": cmd, "kernel": info, "counters_per_launch": out, "failed_passes": failed}
if "SQ_WAVE_CYCLES" in out:
    wc = out["SQ_WAVE_CYCLES"]["mean"]
    res["wave_cycle_shares"] = {"active": out["SQ_ACTIVE_INST_ANY"]["mean"] / wc, "wait_mem": out["SQ_WAIT_ANY"]["mean"] / wc, "wait_issue": out["SQ_WAIT_INST_ANY"]["mean"] / wc}
if "TCC_EA0_RDREQ_sum" in out:
    res["memory_side_bytes_per_launch"] = 2.0 * 64.0 * out["TCC_EA0_RDREQ_sum"]["mean"] + 64.0 * out["TCC_EA0_WRREQ_sum"]["mean"]
json.dump(res, open(outp, "w"), indent=1)
print(json.dumps({k: res[k] for k in res if k != "command"})[:1500])
