"""The 200-video job with shot detection inside (bench.config3_job, mode 'shot_net') with 2 / 3 / 4 planner threads and other look-ahead
bounds (GPU box helper).  argv: 'P[:ahead]' ...   e.g.  python tools/time_shot_job_planners.py 3 2 4 3:8 3:40"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from retargetvid_amd import scheduler, weights
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
sd = weights.make_synthetic_state_dict(0)
_init = scheduler.JobScheduler.__init__
for spec in (sys.argv[1:] or ['3', '2', '4']):
    p, _, ahead = spec.partition(':')
    scheduler.JobScheduler.PLANNERS = int(p)

    def init(self, *a, _ahead=ahead, **kw):
        if _ahead:
            kw['plan_ahead'] = int(_ahead)
        _init(self, *a, **kw)
    scheduler.JobScheduler.__init__ = init
    r = bench.config3_job(1, 0, False, dev, sd, 12, None, mode='shot_net')
    print('planners %s look-ahead %s: %s s  (all runs %s; high water %s)' % (p, ahead or 'default', r['seconds'], r['seconds_all_runs'],
                                                                           r['scheduler_rank0'].get('plan_high_water')), flush=True)
scheduler.JobScheduler.__init__ = _init
