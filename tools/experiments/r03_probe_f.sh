export TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/libtds_hip_x1614p.so
for id in 20 21 22 24 25; do echo "probe $id: $(TDS_GRAM_STAMP_AT=$id python tools/profile_phases.py ant 4096 0 100 2>/dev/null | grep -E 'H LDLt  \(|F solve|not stamped' | tr -s ' ' | tr '\n' '|')"; done
