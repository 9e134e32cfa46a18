for w in 1 2 3 4; do timeout 300 python tools/run_config3.py --workers $w 2>&1 | tail -1 | cut -c1-330; done
