OMDS_EXACT_TALL=1 python tools/exact_timeline.py | tail -16
