for NW in 1 2; do echo "== FAR_WAVES=$NW TW=512"; MONORTM_FAR_WAVES=$NW tools/trace_kernels.sh c3 r05_m_fw$NW | grep "far_kernel\|lines_kernel"; done
for NW in 1 2; do echo "== FAR_WAVES=$NW TW=256"; MONORTM_TILE_WAVES=2 MONORTM_FAR_WAVES=$NW tools/trace_kernels.sh c3 r05_m_fw${NW}_tw2 | grep "far_kernel\|lines_kernel"; done
