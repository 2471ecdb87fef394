# BASELINE config #4 on one GPU (parity mode): JSON into profiles/, kernel statistics of the same command under rocprofv3
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python3 tools/bench_sc_query.py "$@" > gpurun_out/sc_query.json 2> gpurun_out/sc_query.err || { echo failed; tail -20 gpurun_out/sc_query.err; exit 1; }
python3 -c "
import json
r=json.load(open('gpurun_out/sc_query.json'))
print('loops', r['loops_detected'], 'of', r['database']['revisits'], 'revisits; gen', round(r['database']['render_and_describe_s'],1),'s')
print('single', {k:v for k,v in r['single'].items() if k!='roofline'}, r['single']['roofline']['frac'])
print('batched', {k:v for k,v in r['batched'].items() if k not in ('roofline','kernel_ms_total')}, r['batched']['roofline']['frac'])
print('cpu', r['cpu_oracle'])
"
tail -3 gpurun_out/sc_query.err
