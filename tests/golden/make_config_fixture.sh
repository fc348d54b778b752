#!/bin/bash
# Regenerates tests/golden/config_min.tar.gz: the minimal subset of the reference's species-parameter DATA
# (config/, read-only input of the drop-in contract; not source code) needed to run the human, fly, arabidopsis
# (old parameter-file format, short donor window: block size 4) and saccharomyces (3 GC classes, ass_end 0) ab-initio models where /root/reference does not exist (the GPU box).  Run in the build container.
set -e
REF=${REF:-/root/reference}
OUT=$(cd "$(dirname "$0")" && pwd)/config_min.tar.gz
tar -C "$REF" -czf "$OUT" \
    config/species/human/human_parameters.cfg config/species/human/human_exon_probs.pbl \
    config/species/human/human_intron_probs.pbl config/species/human/human_igenic_probs.pbl \
    config/species/human/human_weightmatrix.txt config/species/human/human_utr_probs.pbl \
    config/species/human/human_trans_shadow_partial_utr.pbl \
    config/species/fly/fly_parameters.cfg config/species/fly/fly_exon_probs.pbl \
    config/species/fly/fly_intron_probs.pbl config/species/fly/fly_igenic_probs.pbl \
    config/species/fly/fly_weightmatrix.txt config/species/fly/fly_utr_probs.pbl \
    config/species/arabidopsis/arabidopsis_parameters.cfg config/species/arabidopsis/arabidopsis_exon_probs.pbl \
    config/species/arabidopsis/arabidopsis_intron_probs.pbl config/species/arabidopsis/arabidopsis_igenic_probs.pbl \
    config/species/arabidopsis/arabidopsis_weightmatrix.txt \
    config/species/saccharomyces/saccharomyces_parameters.cfg config/species/saccharomyces/saccharomyces_exon_probs.pbl \
    config/species/saccharomyces/saccharomyces_intron_probs.pbl config/species/saccharomyces/saccharomyces_igenic_probs.pbl \
    config/species/saccharomyces/saccharomyces_weightmatrix.txt \
    config/model config/extrinsic/extrinsic.cfg config/parameters/aug_cmdln_parameters.json
ls -la "$OUT"
