set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python tools/exp_copy.py final 2>/dev/null
python tools/exp_mixed.py final 2>/dev/null
IST_LDS_RUN=1 python tools/exp_mixed.py run1 2>/dev/null
IST_LDS_RUN=3 python tools/exp_mixed.py run3 2>/dev/null
