#!/bin/bash
# round-2 trip B: full GPU suite (new: double, notebook, extension points, large-code state), counter list
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== new tests"; timeout 1500 python -m pytest tests/test_gpu_double.py tests/test_gpu_notebook.py tests/test_gpu_idd.py "tests/test_gpu_parity.py::test_decoder_callbacks_and_custom_updates_on_device" "tests/test_gpu_parity.py::test_c2_boxplus_phi_at_scale_vs_oracle" -q -m gpu 2>&1 | tail -40
echo "== full suite"; timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
echo "== counters"; (rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -o -i -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|INSTS_[A-Z_]*|WAIT[A-Z_]*|BARRIER|LDS[A-Z_]*|ACTIVE[A-Z_]*|BUSY[A-Z_]*|WAVE[A-Z_]*|VALU[A-Z_]*|LEVEL[A-Z_]*))\b" | sort -u | tr '\n' ' ' > gpurun_out/counters_r02b.txt; wc -c gpurun_out/counters_r02b.txt
