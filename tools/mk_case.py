import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from monortm_amd import caseio, synth, tape3
d = sys.argv[1]
rec = synth.synthetic_lines(500); wn = synth.c2_channels(50)
profs = [synth.perturbed_profile(i, wn, nlay=64) for i in range(64)]
tape3.write_tape3(os.path.join(d, "TAPE3"), rec); caseio.write_case(os.path.join(d, "case.bin"), profs)
