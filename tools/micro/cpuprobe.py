import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import model_ref
from yolo_master_amd.nn.tasks import DetectionModel, yaml_model_load
from yolo_master_amd.weights import synth_input, synth_state_dict
print("cores", os.cpu_count(), "torch threads default", torch.get_num_threads())
cfg = yaml_model_load("yolo-master-s.yaml"); sd = synth_state_dict(DetectionModel(cfg).state_dict(), seed=0)
x = synth_input(2, 640, 640, seed=1)
for nt in (8, 16, 32):
    torch.set_num_threads(nt)
    with torch.inference_mode():
        t=time.time(); model_ref.forward(cfg, sd, x); t1=time.time()-t
        t=time.time(); model_ref.forward(cfg, sd, x); t2=time.time()-t
    print(nt, "threads: first %.2fs second %.2fs" % (t1, t2), flush=True)
