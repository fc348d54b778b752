// genes.h -- state path -> transcripts -> genes -> GFF text (host side, after the device decode).
// Restates the reference's post-processing for the ab-initio, non-UTR model:
//   State::setTruncFlag / getBiologicalState      reference src/gene.cc:176-321
//   StatePath::projectOntoGeneSequence            reference src/gene.cc:394-700
//   filterGenePrediction / groupTranscriptsToGenes reference src/gene.cc:2465-2524, 3191-3240
//   printGeneList / Gene::printGFF / printProteinSeq / printEvidence   reference src/gene.cc:3071-3120,1998-2313,2356-2445
#pragma once
#include <string>
#include <vector>
#include "model.h"

namespace augx {

enum { TRUNC_LEFT = 1, TRUNC_RIGHT = 2 };
const int TYPE_INTRON = 71, TYPE_RINTRON = 72; // intron_type, rintron_type (reference include/types.hh:507)

struct BioState {
    long begin = 0, end = 0;
    int type = -1;
    int truncated = 0;
    int framemod = 0;
    // posterior probability of the exon / intron after sampling (reference State::apostprob, a float; include/gene.hh:117-119)
    float apostprob = 0;
    int sampleCount = 1;
    bool hasScore = false;
    int frame() const;
    int length() const { return (int)(end - begin + 1); }
};

struct Transcript {
    std::vector<BioState> exons, introns;
    // untranslated regions (--UTR=on; reference Gene::utr5exons / utr3exons / utr5introns / utr3introns, include/gene.hh:405-416)
    std::vector<BioState> utr5exons, utr3exons, utr5introns, utr3introns;
    bool complete5utr = true, complete3utr = true;
    bool plus = true;
    int frame = 0;
    bool complete = true;
    long transstart = -1, transend = -1, codingstart = -1, codingend = -1;
    int clength = 0;
    std::string id, geneid, seqname;
    // sampling (reference Transcript::apostprob / hasProbs / viterbi / throwaway, include/gene.hh)
    float apostprob = 1.0f;
    bool hasProbs = false, viterbi = true, throwaway = false;
    int serial = 0;      // order of creation within a run: the transcripts of the Viterbi path, then those of the sampled paths (the reference's addresses fall in it)
    bool revRun = false; // mapped back from the run on the reverse complement (--singlestrand=true)
    double meanStateProb() const;
    long geneBegin() const { return transstart >= 0 ? transstart : codingstart; }
    long geneEnd() const { return transend >= 0 ? transend : codingend; }
    bool completeCDS() const;
    void shift(long d);
};

struct GeneOut {
    std::vector<Transcript> transcripts;
    std::string id, seqname;
    bool plus = true;
    long mincodstart = 0, maxcodend = 0;
    float apostprob = 0;
    bool hasProbs = true; // (the gene line's score; reverseGeneList builds its AltGenes afresh, with hasProbs = false: '.')
};

struct OutputOptions {
    bool print_utr = false, print_tss = false, print_tts = false, utr = false;
    bool print_start = true, print_stop = true, print_introns = false, print_cds = true, print_exonnames = false,
         gff3 = false, stopCodonExcludedFromCDS = false, protein = true, codingseq = false, evidence = false,
         uniqueGeneId = false, softmasking = true;
    long offset = 0; // added to every printed coordinate (--predictionStart)
    std::string transTable; // amino acids by codon index (--translation_table; empty: the standard code)
    void fromModel(const Model &m);
};

// raw decoded path of one piece (0-based HMM-state coordinates, 5'->3', chain states merged)
struct PathState { long begin, end; int type; };

// path -> list of (possibly partial) transcripts in piece coordinates
std::vector<Transcript> projectOntoGeneSequence(const Model &m, const std::vector<PathState> &path, long dnalen);
// drop transcripts the reference's filterGenePrediction would drop (ab initio: CDS length rules only)
// (anyStrand: the --strand option is not applied -- runs of the single-strand model, where the caller picks the runs)
// seq: the sequence the run was made on (piece coordinates; the reverse complement of the piece for the second run of the
// single-strand model) -- read with --noInFrameStop=true only, which drops transcripts with a stop codon inside their CDS
std::vector<Transcript> filterTranscripts(const Model &m, const std::vector<Transcript> &txs, bool anyStrand = false, const char *seq = nullptr);
// a transcript predicted on the reverse complement of a piece (--singlestrand=true) mapped back onto the piece: coordinates
// mirrored at endpos = piece length - 1, exon types turned into those of the other strand (reference reverseGeneSequence,
// src/gene.cc:3246-3365)
void reverseTranscript(Transcript &t, long endpos);
// the transcripts of the Viterbi path and of the sampled paths of one piece, identical ones united, with the posterior
// probabilities of transcripts, exons and introns estimated from the sample (reference NAMGene::findGenes, src/namgene.cc:795-905;
// sampleiterations = the Viterbi path + the sampled ones)
std::vector<Transcript> posteriorTranscripts(const Model &m, const std::vector<PathState> &viterbi,
                                             const std::vector<std::vector<PathState>> &samples, long dnalen, int sampleiterations);
// the kept transcripts of one run grouped into genes: --maxtracks, overlapping transcripts of one strand and reading frame as the
// alternatives of one gene (--alternatives-from-sampling=true; one path alone has no overlaps), genes sorted by coding start
std::vector<GeneOut> groupToGenes(const Model &m, const std::vector<Transcript> &txs);
// the genes of a run on the reverse complement mapped back onto the piece (endpos = piece length - 1)
void reverseGenes(std::vector<GeneOut> &genes, long endpos);
// print the genes of one piece.  seq = the WHOLE input sequence (lower/upper case irrelevant), offset-free coordinates.
// rmRuns: the soft-masked runs [first, last] that are hint groups of this piece (evidence block); NULL = every lower-case
// run of seq (a record decoded as one piece)
void printGeneList(std::string &out, const std::vector<GeneOut> &genes, const char *seq, long seqlen, const OutputOptions &o,
                   const std::vector<std::pair<long, long>> *rmRuns = nullptr);

std::string translateCDS(const char *codingSeq, const char *table = nullptr); // (table: 64 letters by codon index, null: the standard code)

} // namespace augx
