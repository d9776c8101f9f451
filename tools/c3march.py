"""config 3's ordered histories, 8 ranks against 1, under the forms of the dominant-pattern product: python tools/c3march.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_config3_gpu as t
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for march in sys.argv[2:] or ["0", "1", "1", "3"]:
    os.environ["CONFIG3_NO_OVERLAP"] = "1" if march.endswith("n") else "0"
    march = march.rstrip("n")
    os.environ["CONFIG3_DOM_MARCH"] = march
    one = t._run(1, N, 6)[0]
    ranks = t._run(8, N, 6)
    print("march", march, "no overlap" if os.environ["CONFIG3_NO_OVERLAP"] == "1" else "", "single", one["ref_hist"]["rhistory"][:4], flush=True)
    for o in ranks[:3]:
        print("   rank", o["rank"], o["ref_hist"]["rhistory"][:4], "product slices equal:", o["y_sha256"][0] == one["y_sha256"][o["rank"]], flush=True)
