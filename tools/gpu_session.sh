O=gpurun_out/r06h; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|error|Error" | tail -8) > $O/pytest_all.log; cat $O/pytest_all.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke) > $O/smoke.log; cat $O/smoke.log
