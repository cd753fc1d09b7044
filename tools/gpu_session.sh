O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rs -rf 2>&1 | grep -v "^loading\|^n_layers\|^n_embed\|amdgpu.ids\|socket.cpp\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|pipeline transport up\|^D=" | tail -40 > $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 900 python bench.py 2>$O/bench7b_full.err | tail -1 > $O/bench7b_full.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/bench7b_full.json'))
print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['prefill']['ms_per_chunk'], d['prefill']['long_prompt']['tokens_per_s'], d['parity_gates_failed'])
PY
