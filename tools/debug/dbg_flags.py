import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from hip_env import HipEnv
import oracle_lib as O
from gym_continuousdoubleauction_amd import _capi as K
n,a,steps=4096,4,200
cfg={"num_of_agents":a,"init_cash":1000000,"max_step":100000,"is_render":False}
env=HipEnv(cfg,n); env.reset(np.arange(5000,5000+n,dtype=np.uint64))
rng=np.random.default_rng(7)
def acts():
    return (rng.integers(0,9,(n,a)).astype(np.int32),rng.uniform(-1,1,(n,a)).astype(np.float32),rng.uniform(0,1,(n,a)).astype(np.float32),rng.integers(0,10,(n,a)).astype(np.int32),rng.integers(0,3,(n,a)).astype(np.int32))
first=None
for t in range(steps):
    env.step(*acts())
    f=env.flags()
    if f.any() and first is None:
        first=t; idx=np.nonzero(f)[0]; print("step",t,"markets",idx[:5],"flags",f[idx[:5]])
        s=env.get_state(int(idx[0]))
        for tr in range(a):
            acc=s.acc[tr]
            print(tr,[str(K.dec_to_decimal(getattr(acc,x))) for x in ("cash","cash_on_hold","position_val","vwap","nav","prev_nav","max_nav")],acc.net_position)
        break
