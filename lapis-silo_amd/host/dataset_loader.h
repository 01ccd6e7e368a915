// dataset_loader.h — reads the reference's *input* files straight into the device store.
//
// Replaces, for this path only, the DuckDB-driven Preprocessor (reference src/silo/preprocessing/
// preprocessor.cpp:36-85, :87-131 ndjson, :255-334 metadata + FASTA; config readers
// preprocessing_config_reader.cpp:18-37, database_config.cpp:47-90; reference_genomes.cpp): no SQL, no
// partitioning — one DatabasePartition with rows in input-file order (the reference orders rows by
// partitionBy / dateToSortBy / primaryKey through DuckDB; counts, Mutations tables and every result of this
// path are independent of row order).
//
// Files understood, all relative to the data set directory:
//   preprocessing_config.yaml   optional; metadataFilename | ndjsonInputFilename, pangoLineageDefinitionFilename,
//                               referenceGenomeFilename, nucleotideSequencePrefix ("nuc_"), genePrefix ("gene_"),
//                               unalignedNucleotideSequencePrefix ("unaligned_")
//   database_config.yaml        schema.primaryKey, schema.dateToSortBy, schema.metadata[] (name / type / generateIndex:
//                               every column is loaded with its reference type), defaultNucleotideSequence
//   reference_genomes.json, pangolineage_alias.json
//   <metadata>.tsv + <prefix><name>.fasta[.zst|.xz] (+ unaligned_<name>.fasta[.zst|.xz])
//   or <input>.ndjson[.zst|.xz] with metadata, aligned / unaligned sequences and the insertion maps per record
#pragma once

#include <string>
#include <vector>

#include "database.h"

namespace silo::preprocessing {

class PreprocessingException : public std::runtime_error {
  public:
   using std::runtime_error::runtime_error;
};

struct DatasetSummary {
   size_t sequence_count = 0;
   size_t nucleotide_stores = 0;
   size_t amino_acid_stores = 0;
   size_t lineage_columns = 0;
   size_t null_sequences = 0;
};

/// Reads (DatabaseConfigReader::readConfig, database_config.cpp:197-231) and, with `validate`, checks
/// (ConfigRepository::validateConfig, config_repository.cpp:22-108) a database_config.yaml and renders
/// what was understood as JSON: {"instanceName", "primaryKey", "dateToSortBy", "partitionBy", "metadata": [{"name", "type",
/// "generateIndex"}]}.  Throws PreprocessingException with the reference's message for an invalid config.
std::string describeDatabaseConfig(const std::string& path, bool validate);

/// The records of a FASTA file (plain / .zst / .xz) as the loader reads them, a JSON array of [key, genome] sorted by key;
/// throws PreprocessingException with the reference's FastaFormatException messages (fasta_reader.cpp:11-47).
std::string describeFasta(const std::string& path);

/// The row order of the loader = the reference's: rows by (partitionBy key, dateToSortBy, primary key), rows without a date last
/// (preprocessor.cpp:159-227, database_config.cpp:190-198); an empty vector = the schema names no such column.  order[k] = the
/// input row that becomes row k.
std::vector<uint32_t> referenceRowOrder(
   const std::vector<std::string>& partition_keys, const std::vector<std::string>& dates, const std::vector<std::string>& primary_keys
);

/// Fills `database` (must be empty) from the files in `directory` and finalises it.
DatasetSummary loadDataset(Database& database, const std::string& directory);

}  // namespace silo::preprocessing
