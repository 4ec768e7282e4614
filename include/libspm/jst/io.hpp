// libspm/jst/io.hpp -- FASTA (plain or .gz) and VCF readers for the journaled-sequence tree.
// Inputs of the kind the reference's test data holds (/root/reference/test/data/sim_ref_10Kb.fasta.gz,
// sim_ref_10Kb_SNPs.vcf, sim_ref_10Kb_SNP_INDELs.vcf; registered in test/data/datasources.cmake:70-102).
// Needs zlib (-lz).
#pragma once

#include <zlib.h>

#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <libspm/seqan/alphabet.hpp>

namespace spm::io
{
inline std::string read_file(std::string const & path) // transparently gunzips
{
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("cannot open " + path);
    std::string out;
    char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof(buf))) > 0)
        out.append(buf, static_cast<std::size_t>(n));
    gzclose(f);
    return out;
}

struct fasta_record
{
    std::string id;
    std::vector<std::uint8_t> ranks; // dna4 ranks (IUPAC codes fold as seqan3::dna4 does)
};

inline std::vector<fasta_record> read_fasta(std::string const & path)
{
    std::vector<fasta_record> recs;
    std::istringstream in{read_file(path)};
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r')
            line.pop_back();
        if (line.empty())
            continue;
        if (line[0] == '>') {
            recs.push_back({line.substr(1), {}});
        } else if (!recs.empty()) {
            for (char c : line)
                recs.back().ranks.push_back(spm::dna4::char_to_rank(c));
        }
    }
    return recs;
}

// One alternative allele of one VCF record, normalised: replace reference [pos, pos + ref_len) by `alt`, with the
// common leading bases of REF and ALT trimmed (so the anchor base of an indel stays free for a SNP at the same POS).
struct vcf_allele
{
    std::size_t pos{};     // 0-based reference position
    std::size_t ref_len{}; // 0 for a pure insertion
    std::vector<std::uint8_t> alt;
    std::vector<std::uint8_t> coverage; // one byte per haplotype: 1 if the haplotype carries this allele
};

struct vcf_data
{
    std::size_t n_haplotypes{};
    std::vector<vcf_allele> alleles; // sorted by pos
};

inline vcf_data read_vcf(std::string const & path)
{
    vcf_data out;
    std::istringstream in{read_file(path)};
    std::string line;
    auto split = [](std::string const & s, char sep) {
        std::vector<std::string> f;
        std::size_t a = 0;
        for (;;) {
            std::size_t b = s.find(sep, a);
            f.push_back(s.substr(a, b == std::string::npos ? b : b - a));
            if (b == std::string::npos)
                break;
            a = b + 1;
        }
        return f;
    };
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') {
            if (line.rfind("#CHROM", 0) == 0)
                out.n_haplotypes = 2 * (split(line, '\t').size() - 9);
            continue;
        }
        auto f = split(line, '\t');
        if (f.size() < 10)
            continue;
        std::size_t const pos = std::stoul(f[1]) - 1;
        std::string const & ref = f[3];
        auto alts = split(f[4], ',');
        std::vector<vcf_allele> rec(alts.size());
        for (std::size_t a = 0; a < alts.size(); ++a) {
            std::string const & alt = alts[a];
            if (!alt.empty() && alt[0] == '<')
                throw std::runtime_error("symbolic ALT alleles are not supported: " + alt);
            std::size_t lead = 0;
            while (lead < ref.size() && lead < alt.size() && ref[lead] == alt[lead])
                ++lead;
            rec[a].pos = pos + lead;
            rec[a].ref_len = ref.size() - lead;
            for (std::size_t i = lead; i < alt.size(); ++i)
                rec[a].alt.push_back(spm::dna4::char_to_rank(alt[i]));
            rec[a].coverage.assign(out.n_haplotypes, 0);
        }
        for (std::size_t s = 9; s < f.size(); ++s) {
            std::string const & gt = f[s];
            std::size_t bar = gt.find_first_of("|/");
            int const a0 = std::stoi(gt.substr(0, bar));
            int const a1 = bar == std::string::npos ? a0 : std::stoi(gt.substr(bar + 1));
            std::size_t const h = 2 * (s - 9);
            if (a0 > 0)
                rec[static_cast<std::size_t>(a0 - 1)].coverage[h] = 1;
            if (a1 > 0 && h + 1 < out.n_haplotypes)
                rec[static_cast<std::size_t>(a1 - 1)].coverage[h + 1] = 1;
        }
        for (auto & r : rec)
            out.alleles.push_back(std::move(r));
    }
    std::stable_sort(out.alleles.begin(), out.alleles.end(),
                     [](vcf_allele const & a, vcf_allele const & b) { return a.pos < b.pos; });
    return out;
}
} // namespace spm::io
