import sys, tempfile, collections
sys.path.insert(0, '.')
from rumpy_amd.shared_framework.models import define_model
h = define_model('edsr', model_save_dir=tempfile.mkdtemp(), device=0, eval_mode=True, scale=4)
h.net._ensure_engine()
eng = h.net.engine
plan = eng.plan_for(1, 339, 510, False, eng.eval_fmt)
print('eval_fmt', eng.eval_fmt)
for op, a in plan.fwd:
    extra = ''
    for k in ('N', 'H', 'W', 'cin_chunks', 'cout_tiles', 'out_mode', 'fmt', 'nblocks', 'col_tile', 'grid_x'):
        if hasattr(a, k):
            extra += ' %s=%s' % (k, getattr(a, k))
    print(op, extra)
