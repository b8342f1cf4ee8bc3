cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_share8; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --simulate-world 8 --batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-isolated --no-extra-schedules > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_share8/kt/*/*_kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f))]
ks=sorted([(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:24],r['Stream_Id']) for r in rows])
gens=[k for k in ks if 'k_generate' in k[2]]
fstart=[g[0] for g in gens][::3]
for fi in range(len(fstart)-2,len(fstart)):
    a=fstart[fi]; b=fstart[fi+1] if fi+1<len(fstart) else ks[-1][1]+1
    fr=[k for k in ks if a<=k[0]<b]
    end=max(k[1] for k in fr)
    iv=sorted((k[0],k[1]) for k in fr); cov=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: cov+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    cov+=ce-cs
    print('frame',fi,'kernels',len(fr),'span us',round((end-a)/1000,1),'busy us',round(cov/1000,1),'idle us',round((end-a-cov)/1000,1))
    # one stream's chain
    s0=fr[0][3]; ch=[k for k in fr if k[3]==s0]
    print(' stream',s0,[ (k[2][:12], round((k[1]-k[0])/1000,1)) for k in ch][:24])
    print(' gaps', [round((ch[i+1][0]-ch[i][1])/1000,1) for i in range(len(ch)-1)][:24])
PY
tail -2 $OUT/bench.log | cut -c1-200
find $OUT -name "*.csv" -size +2M -delete
