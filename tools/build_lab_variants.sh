# The four lab libraries of the bound-evidence experiment (tools/collect_bound_evidence.sh), built HERE before the gpurun call:
#   nori_amd/lib/libnori_hip_lab_{base,valu,load,idle}.so   (wf_experiments.h: NORI_EXP_SENS / NORI_EXP_WIDE_SENS = none / 3 / 1 / 5)
set -e
D=$(cd $(dirname $0) && pwd)
bash $D/build_variant_fast.sh lab_base
bash $D/build_variant_fast.sh lab_valu -DNORI_EXP_SENS=3 -DNORI_EXP_WIDE_SENS=3
bash $D/build_variant_fast.sh lab_load -DNORI_EXP_SENS=1 -DNORI_EXP_WIDE_SENS=1
bash $D/build_variant_fast.sh lab_idle -DNORI_EXP_SENS=5 -DNORI_EXP_WIDE_SENS=5
