cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for order in "0 1 2 3 4 5 6 7" "7 6 5 4 3 2 1 0" "0 3 1 4 0 3"; do
  rm -rf gpurun_out/qq; timeout -k 5 60 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qq -o r -- ./tools/exp/queues $order > /dev/null 2>&1
  echo "order: $order"; python3 - <<'P'
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/qq/**/*kernel_trace.csv",recursive=True)[0])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
print(" ".join("s%d->q%s"%(int(r.get("Grid_Size_X",r.get("Grid_Size")))//64-1,r["Queue_Id"]) for r in rows))
P
done
