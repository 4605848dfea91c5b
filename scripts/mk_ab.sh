# build build_ab/libA.so from the committed sources and build_ab/libB.so from the working tree (for scripts/ab_lib.sh)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
L=multimodal-model-skin-lesion-classifier_amd/libmmskin_hip.so
make -C multimodal-model-skin-lesion-classifier_amd/csrc -j6 >/dev/null && cp $L build_ab/libB.so
git stash -q && make -C multimodal-model-skin-lesion-classifier_amd/csrc -j6 >/dev/null && cp $L build_ab/libA.so; git stash pop -q
make -C multimodal-model-skin-lesion-classifier_amd/csrc -j6 >/dev/null
ls -la build_ab
