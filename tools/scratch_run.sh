for NS in 3 5; do echo "== NSLICE=$NS"; MONORTM_NSLICE=$NS tools/trace_kernels.sh c3 r05_q_ns$NS | grep "lines_kernel\|reduce_slices"; done
for FL in 3; do echo "== FAR_LEVELS=$FL"; MONORTM_FAR_LEVELS=$FL tools/trace_kernels.sh c3 r05_q_fl$FL | grep "lines_kernel\|far_"; done
