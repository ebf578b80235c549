import sys
args = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
sys.argv = [sys.argv[0]]
sys.path.insert(0, '/root/repo/tools')
import exp_r2a as E
gpt = E.build()
E.e2(gpt, args or [(64, 1), (96, 1), (128, 1), (64, 2), (48, 2)])
