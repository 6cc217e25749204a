import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mv2d_amd import configs, registry, synthetic
import mv2d_amd.plugin
dev='cuda'
prob = synthetic.make_problem('cfg2_s', seed=0)
cfg = configs.roi_head_cfg_s()
head = registry.build_head(cfg, train_cfg=configs.TRAIN_CFG_RCNN, test_cfg=configs.TEST_CFG_RCNN)
head.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_head_state(seed=0).items()}, strict=False)
head = head.to(dev)
gtc = synthetic.make_train_gt(40, 3)
gt, labels = [torch.from_numpy(gtc['gt'])], [torch.from_numpy(gtc['gt_labels'])]
feat = torch.from_numpy(prob['feat']).to(dev).requires_grad_(True)
props = [torch.from_numpy(p) for p in prob['proposals']]
metas = [dict(m, box_type_3d=None) for m in prob['img_metas']]
def fwd():
    return head.forward_train([feat], metas, props, None, None, None, None, gt, labels, None, autograd=True)
def full():
    l = fwd()
    for p in head.parameters(): p.grad = None
    feat.grad = None
    sum(l.values()).backward()
for _ in range(3): full()
torch.cuda.synchronize()
# wall time of forward host-only (no sync) vs synced
t0=time.perf_counter(); l=fwd(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('forward host %.2f ms, +sync %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3))
t0=time.perf_counter(); sum(l.values()).backward(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('backward host %.2f ms, +sync %.2f ms' % ((t1-t0)*1e3, (t2-t1)*1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): fwd()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative'); st.print_stats(45)
# kernel count per step
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    full(); torch.cuda.synchronize()
ka = p.key_averages()
tot_k = sum(e.count for e in ka if e.device_type.name == 'CUDA' or getattr(e,'self_device_time_total',0) > 0)
print('events with device time:', tot_k)
print(ka.table(sort_by='self_cpu_time_total', row_limit=25, max_name_column_width=50))
print(ka.table(sort_by='self_device_time_total', row_limit=22, max_name_column_width=70))
