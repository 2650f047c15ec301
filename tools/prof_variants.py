"""Average kernel durations of the illumination kernels under TBRM_DEBUG variants (diagnostic; timing experiments only)."""
import csv, os, subprocess, sys, shutil
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
variants = [int(v) for v in sys.argv[1:]] or [0, 16, 32, 48, 256]
for d in variants:
    shutil.rmtree("/tmp/po", ignore_errors=True); os.makedirs("/tmp/po")
    env = dict(os.environ, TBRM_DEBUG=str(d), TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", "/tmp/po", "-o", "p", "--output-format", "csv", "--",
                    sys.executable, os.path.join(R, "tools", "prof_light.py")], cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    print("debug", d)
    for row in csv.DictReader(open("/tmp/po/p_kernel_stats.csv")):
        if "k_light" in row["Name"] or "k_occ_flags" in row["Name"]:
            print(f"   {row['Name'][11:58]:48s} calls {row['Calls']:>5s}  avg {float(row['AverageNs'])/1000:8.2f} us")
