// query_engine.cpp — Query, QueryEngine::executeQuery and the result JSON.
// Reference: src/silo/query_engine/{query,query_engine,query_result}.cpp.
#include "query_engine.h"

#include <chrono>

namespace silo::query_engine {

Query::Query(const std::string& query_string) {  // query.cpp:13-28
   json::Value json;
   try {
      json = json::parse(query_string);
   } catch (const json::ParseError& ex) {
      throw QueryParseException("The query was not a valid JSON: " + std::string(ex.what()));
   }
   if (!json.contains("filterExpression") || !json["filterExpression"].is_object() || !json.contains("action") ||
       !json["action"].is_object()) {
      throw QueryParseException("Query json must contain filterExpression and action.");
   }
   try {
      filter = filter_expressions::parseExpression(json["filterExpression"]);
      action = actions::parseAction(json["action"]);
   } catch (const std::out_of_range& ex) {
      // nlohmann raises json::exception for wrong-typed accesses, mapped to 400 (query.cpp:24-27)
      throw QueryParseException("The query was not a valid JSON: " + std::string(ex.what()));
   }
}

namespace {

/// Microseconds `work` takes, as the reference's BlockTimer reports them (block_timer.h:5-23, query_engine.cpp:63-65).
template <typename Work>
int64_t microsecondsOf(Work&& work) {
   const auto begin = std::chrono::steady_clock::now();
   work();
   return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - begin).count();
}

/// Expression::compile + Operator::evaluate per partition (query_engine.cpp:40-49).  evaluate() is lazy here: the fused
/// kernel is launched by the action, in the mode it needs (count only for Aggregated, bitset + count for Mutations).
std::vector<OperatorResult> compileFilter(const Database& database, const filter_expressions::Expression& filter) {
   std::vector<OperatorResult> per_partition;
   per_partition.reserve(database.partitions.size());
   for (const DatabasePartition& partition : database.partitions) {
      per_partition.push_back(operators::Operator::evaluate(filter.compile(database, partition, filter_expressions::Expression::AmbiguityMode::NONE)));
   }
   return per_partition;
}

}  // namespace

QueryResult QueryEngine::executeQuery(const std::string& query_string) const {  // query_engine.cpp:30-68
   Trace::reset();
   const Query query(query_string);
   Trace::mark("parsed");
   std::vector<OperatorResult> filters;
   const int64_t filter_time = microsecondsOf([&] { filters = compileFilter(database, *query.filter); });
   Trace::mark("compiled");
   QueryResult rows;
   const int64_t action_time = microsecondsOf([&] { rows = query.action->executeAndOrder(database, std::move(filters)); });
   Trace::mark("action_done");
   Database::lastTimings() = {filter_time, action_time};
   return rows;
}

std::vector<QueryEngine::BatchOutcome> QueryEngine::executeQueries(const std::vector<std::string>& queries) const {
   Trace::reset();
   std::vector<BatchOutcome> outcomes(queries.size());
   std::vector<std::unique_ptr<Query>> parsed(queries.size());
   std::vector<std::unique_ptr<actions::Action::Pending>> pending(queries.size());
   actions::ScanBatcher batcher;  // active on this thread until the end of the function
   for (size_t i = 0; i < queries.size(); ++i) {  // phase 1: parse, compile, evaluate filters, queue scans
      const actions::ScanBatcher::Checkpoint mark = batcher.checkpoint();
      try {
         parsed[i] = std::make_unique<Query>(queries[i]);
         pending[i] = parsed[i]->action->begin(database, compileFilter(database, *parsed[i]->filter));
      } catch (...) {
         batcher.rollback(mark);  // scans recorded by the failed query point into buffers that are gone
         outcomes[i].error = std::current_exception();
      }
   }
   Trace::mark("batch_compiled");
   // The filter -> count queries of the batch (Aggregated without groupByFields): every filter program of a partition
   // in ONE launch (K3b) instead of one latency-bound launch per query.
   struct CountJob {
      size_t query;
      size_t partition;
      std::unique_ptr<ProgramBuilder> builder;  // owns the code and leaf arrays the program points into
      silo_gpu_bitprog program;
   };
   std::vector<CountJob> jobs;
   for (size_t i = 0; i < queries.size(); ++i) {
      if (outcomes[i].error != nullptr || !parsed[i]->action->countsOnly()) {
         continue;
      }
      const size_t first_job = jobs.size();
      try {
         for (size_t partition_index = 0; partition_index < pending[i]->bitmap_filter.size(); ++partition_index) {
            const OperatorResult& filter = pending[i]->bitmap_filter[partition_index];
            CountJob job{i, partition_index, std::make_unique<ProgramBuilder>(filter.rows()), {}};
            if (filter.prepareCount(*job.builder, job.program)) {
               jobs.push_back(std::move(job));
            }
         }
      } catch (...) {
         jobs.resize(first_job);
         outcomes[i].error = std::current_exception();
      }
   }
   Trace::mark("batch_lowered");
   if (jobs.size() > 1) {
      for (size_t partition_index = 0; partition_index < database.partitions.size(); ++partition_index) {
         std::vector<silo_gpu_bitprog> programs;
         std::vector<const CountJob*> members;
         for (const CountJob& job : jobs) {
            if (job.partition == partition_index) {
               programs.push_back(job.program);
               members.push_back(&job);
            }
         }
         if (programs.empty()) {
            continue;
         }
         std::vector<uint64_t> counts(programs.size(), 0);
         checkGpu(
            silo_gpu_filter_eval_batch(
               database.partitions[partition_index].store, programs.data(), static_cast<uint32_t>(programs.size()), nullptr, counts.data(), queryStream()
            ),
            "silo_gpu_filter_eval_batch"
         );
         for (size_t k = 0; k < members.size(); ++k) {
            pending[members[k]->query]->bitmap_filter[partition_index].setCount(static_cast<uint32_t>(counts[k]));
         }
      }
   }
   jobs.clear();  // a single job takes the ordinary path in finish(): one launch with the count slot
   Trace::mark("batch_queued");
   batcher.flush();  // the scans of all queries, several filters per pass over the planes
   Trace::mark("batch_launched");
   for (size_t i = 0; i < queries.size(); ++i) {  // phase 2: fetch, build rows, order
      if (outcomes[i].error != nullptr) {
         continue;
      }
      try {
         outcomes[i].result = parsed[i]->action->finish(database, *pending[i]);
      } catch (...) {
         outcomes[i].error = std::current_exception();
      }
   }
   Trace::mark("batch_done");
   return outcomes;
}

json::Value toJson(const QueryResult& query_result) {  // query_result.cpp:10-25
   json::Value rows = json::Value::array();
   for (const auto& entry : query_result.query_result) {
      json::Value row = json::Value::object();
      for (const auto& [field, value] : entry.fields) {
         if (!value.has_value()) {
            row.set(field, json::Value());
         } else if (std::holds_alternative<std::string>(*value)) {
            row.set(field, json::Value(std::get<std::string>(*value)));
         } else if (std::holds_alternative<int32_t>(*value)) {
            row.set(field, json::Value(std::get<int32_t>(*value)));
         } else {
            row.set(field, json::Value(std::get<double>(*value)));
         }
      }
      rows.push_back(std::move(row));
   }
   json::Value out = json::Value::object();
   out.set("queryResult", std::move(rows));
   return out;
}

std::string toJsonText(const QueryResult& query_result) {
   // same bytes as toJson(query_result).dump(), written straight from the rows (a Mutations response has
   // hundreds of rows; the Value tree costs more than the text)
   std::string out;
   out.reserve(32 + query_result.query_result.size() * 96);
   out += "{\"queryResult\":[";
   bool first_row = true;
   for (const auto& entry : query_result.query_result) {
      if (!first_row) {
         out.push_back(',');
      }
      first_row = false;
      out.push_back('{');
      bool first_field = true;
      for (const auto& [field, value] : entry.fields) {
         if (!first_field) {
            out.push_back(',');
         }
         first_field = false;
         json::Value::appendString(out, field);
         out.push_back(':');
         if (!value.has_value()) {
            out += "null";
         } else if (std::holds_alternative<std::string>(*value)) {
            json::Value::appendString(out, std::get<std::string>(*value));
         } else if (std::holds_alternative<int32_t>(*value)) {
            out += std::to_string(std::get<int32_t>(*value));
         } else {
            json::Value::appendDouble(out, std::get<double>(*value));
         }
      }
      out.push_back('}');
   }
   out += "]}";
   return out;
}

}  // namespace silo::query_engine
