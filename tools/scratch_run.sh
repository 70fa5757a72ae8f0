for NS in 1 2 3 5; do echo "== NSLICE=$NS"; MONORTM_NSLICE=$NS tools/ab_libs.sh c3 - | tail -1; done
