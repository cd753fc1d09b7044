for i in 1 2; do NO_TL=1 STEPS=1024 bash tools/quick_gpu.sh base g4 g16 2>&1 | grep -v "^L=\|^loading"; done > gpurun_out/graph_tokens_ab.txt; cat gpurun_out/graph_tokens_ab.txt
