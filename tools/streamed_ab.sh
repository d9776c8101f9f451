cd $GRAFT_REPO_ROOT
{
echo "== dom_probe, default placement"; DOM_FORMS=none python tools/dom_probe.py 512 2 2>&1 | grep "values streamed\|addresses"
echo "== bench"; python bench.py --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d['values_streamed']; print('bench values_streamed kernel_ms', s['kernel_ms'], 'ms/step', s['ms_per_step'], 'nontrivial', s['nontrivial_x']['kernel_ms'], 'headline', d['roofline']['kernel_ms'])"
for o in "X_OFF=4096" "X_OFF=65536 Y_OFF=131072" "V_OFF=1048576" "X_OFF=2097152 Y_OFF=4194304 V_OFF=6291456" "X_OFF=1024 Y_OFF=2048 V_OFF=512"; do echo "== dom_probe $o"; env $o DOM_FORMS=none python tools/dom_probe.py 512 2 2>&1 | grep "values streamed\|addresses"; done
echo "== dom_probe again, default"; DOM_FORMS=none python tools/dom_probe.py 512 2 2>&1 | grep "values streamed\|addresses"
} > gpurun_out/streamed_ab.txt 2>&1
cat gpurun_out/streamed_ab.txt
