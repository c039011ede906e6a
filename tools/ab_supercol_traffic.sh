export TMPDIR=/tmp
for sn in 3 4 6 0; do
  rm -rf gpurun_out/pmc_sn${sn}
  VH_PP_SN=$sn timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_sn${sn} -- python3 bench.py --no-cpu-baseline --no-parity --no-fp16-line --steps 3 --warmup 1 > gpurun_out/pmc_sn${sn}.log 2>&1
  python3 - <<PY
import csv,glob,collections
tot=collections.defaultdict(lambda:[0.0,0]); dur=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("gpurun_out/pmc_sn${sn}/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for tag,name in (("BF16, 6","fc1"),("BF16, 5","qkv"),("BF16, 8","proj/fc2")):
            if "gemm_nt_pp_kernel" in k and tag in k and r["Counter_Name"]=="FETCH_SIZE":
                tot[name][0]+=float(r["Counter_Value"]); tot[name][1]+=1
for f in glob.glob("gpurun_out/pmc_sn${sn}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for tag,name in (("BF16, 6","fc1"),("BF16, 5","qkv"),("BF16, 8","proj/fc2")):
            if "gemm_nt_pp_kernel" in k and tag in k:
                dur[name][0]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3; dur[name][1]+=1
print("sn=${sn}", {n:(round(2*tot[n][0]/tot[n][1]*1024/1e6,1), round(dur[n][0]/dur[n][1],1)) for n in tot}, "(MB read x2-corrected, us)")
PY
done
