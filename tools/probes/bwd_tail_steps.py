import os, sys
ROOT='/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT','.')
sys.path.insert(0, os.path.join(ROOT,'retinanet-tensorflow2.x_amd')); sys.path.insert(0, ROOT)
import torch
from retinanet.cfg import default_params
from retinanet.model import ModelBuilder
from retinanet.model.train_engine import TrainEngine
dev=torch.device('cuda:0')
p=default_params(input_size=640, batch_train=32)
b=ModelBuilder(p,'train',device=dev,seed=1337); m=b()
rx=[b.FREEZE_VARS_REGEX[n] for n in p.training.freeze_variables]
eng=TrainEngine(m,32,frozen_regexes=rx,world_size=1)
steps=eng.bwd_steps
last_main=max(i for i,f in enumerate(steps) if not getattr(f,'side',False))
print('steps',len(steps),'last main',last_main)
for i in range(max(0,len(steps)-30),len(steps)):
    f=steps[i]
    print(i,'side' if getattr(f,'side',False) else 'MAIN', getattr(f,'writes',[])[:3], f.__name__ if hasattr(f,'__name__') else '')
