import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import reference_replay as rr
lib = sys.argv[1]
for name in ('default118_soft', 'default118_tight_soft', 'default118_wild_soft', 'default118_hard'):
    try:
        c = rr.replay_engine(lib, name, batch=3)
        print(lib, name, 'OK', {k: c[k] for k in ('steps', 'obs', 'islands')})
    except AssertionError as e:
        print(lib, name, 'FAILED:', str(e)[:300].replace('\n', ' '))
