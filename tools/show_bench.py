"""Pretty-print a bench.py JSON line: python tools/show_bench.py file.json [max_depth]"""
import json
import sys


def show(o, ind=0, maxd=3):
    for k, v in o.items():
        if isinstance(v, dict) and ind // 2 < maxd:
            print(" " * ind + k + ":")
            show(v, ind + 2, maxd)
        else:
            print(" " * ind + k + ": " + str(v)[:150])


lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
show(json.loads(lines[-1]), 0, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
