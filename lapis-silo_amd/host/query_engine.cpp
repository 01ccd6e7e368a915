// query_engine.cpp — Query, QueryEngine::executeQuery and the result JSON.
// Reference: src/silo/query_engine/{query,query_engine,query_result}.cpp.
#include "query_engine.h"

#include <unordered_map>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

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


/// A few worker threads for the host side of a batch: parsing and compiling its queries are independent of each other
/// (≈ 20 µs each for a 32-leaf filter, against 7 µs of device time per query in a multi-program launch), so a batch of 64
/// would otherwise spend three times as long on the host as on the device.  The workers only parse and compile — anything
/// that records into the calling thread's ScanBatcher stays on the calling thread.  Never joined (process exit may come
/// after the runtime libraries have shut down); idle workers sleep on a condition variable.
class BatchWorkers {
  public:
   static BatchWorkers& instance() {
      static BatchWorkers* workers = new BatchWorkers();  // intentionally leaked
      return *workers;
   }

   /// Runs task(i) for i in [0, n) on the workers and the calling thread; returns when all are done.
   template <typename Task>
   void run(size_t n, Task&& task) {
      if (n < 4 || threads.empty()) {
         for (size_t i = 0; i < n; ++i) {
            task(i);
         }
         return;
      }
      Job job;
      job.n = n;
      job.task = [&task](size_t i) { task(i); };
      {
         const std::lock_guard<std::mutex> lock(mutex);
         jobs.push_back(&job);
      }
      wake.notify_all();
      work(job);  // the caller takes its share
      std::unique_lock<std::mutex> lock(mutex);
      jobs.erase(std::remove(jobs.begin(), jobs.end(), &job), jobs.end());
      done.wait(lock, [&job] { return job.finished.load() == job.n && job.active.load() == 0; });
   }

  private:
   struct Job {
      size_t n = 0;
      std::function<void(size_t)> task;
      std::atomic<size_t> next{0};
      std::atomic<size_t> finished{0};
      std::atomic<int> active{0};  // workers currently inside work(job)
   };

   BatchWorkers() {
      const unsigned hardware = std::thread::hardware_concurrency();
      const unsigned count = std::min(7u, hardware > 2 ? hardware / 2 - 1 : 0u);
      for (unsigned k = 0; k < count; ++k) {
         threads.emplace_back([this] { loop(); });
         threads.back().detach();
      }
   }

   void work(Job& job) {
      for (size_t i = job.next.fetch_add(1); i < job.n; i = job.next.fetch_add(1)) {
         job.task(i);  // tasks catch their own exceptions
         job.finished.fetch_add(1);
      }
   }

   void loop() {
      std::unique_lock<std::mutex> lock(mutex);
      while (true) {
         Job* job = nullptr;
         for (Job* candidate : jobs) {
            if (candidate->next.load() < candidate->n) {
               job = candidate;
               break;
            }
         }
         if (job == nullptr) {
            wake.wait(lock);
            continue;
         }
         job->active.fetch_add(1);
         lock.unlock();
         work(*job);
         lock.lock();
         job->active.fetch_sub(1);
         done.notify_all();
      }
   }

   std::mutex mutex;
   std::condition_variable wake;
   std::condition_variable done;
   std::vector<Job*> jobs;
   std::vector<std::thread> threads;
};

}  // namespace

namespace {

/// Parsed queries of the calling thread by their text.  A query text maps to one Expression / Action pair whatever the
/// database holds (names are resolved by compile, per partition), both are immutable once parsed, and dashboards send the same
/// few texts over and over: a repeated query skips the JSON parse (11 of the 45 us of a filter -> count query).  Per thread: no
/// lock on the request path; bounded by dropping everything when full.  Texts that do not parse are not kept.
std::shared_ptr<const Query> parsedQuery(const std::string& query_string) {
   constexpr size_t MAX_ENTRIES = 256;
   constexpr size_t MAX_TEXT_BYTES = 8192;
   thread_local std::unordered_map<std::string, std::shared_ptr<const Query>> cache;
   if (query_string.size() > MAX_TEXT_BYTES) {
      return std::make_shared<const Query>(query_string);
   }
   const auto found = cache.find(query_string);
   if (found != cache.end()) {
      return found->second;
   }
   auto query = std::make_shared<const Query>(query_string);
   if (cache.size() >= MAX_ENTRIES) {
      cache.clear();
   }
   cache.emplace(query_string, query);
   return query;
}

}  // namespace

QueryResult QueryEngine::executeQuery(const std::string& query_string) const {  // query_engine.cpp:30-68
   Trace::reset();
   checkGpu(silo_gpu_set_device(database.device), "silo_gpu_set_device");  // HIP's current device is per thread (request threads start at 0)
   Database::queryFingerprint() = database.all_reduce != nullptr ? Database::fingerprintOf(query_string) : 0;  // only collectives carry it
   const std::shared_ptr<const Query> cached = parsedQuery(query_string);
   const Query& query = *cached;
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

std::string QueryEngine::executeQueryJson(const std::string& query_string) const {
   Trace::reset();
   checkGpu(silo_gpu_set_device(database.device), "silo_gpu_set_device");
   Database::queryFingerprint() = database.all_reduce != nullptr ? Database::fingerprintOf(query_string) : 0;
   const std::shared_ptr<const Query> cached = parsedQuery(query_string);
   const Query& query = *cached;
   Trace::mark("parsed");
   std::vector<OperatorResult> filters;
   const int64_t filter_time = microsecondsOf([&] { filters = compileFilter(database, *query.filter); });
   Trace::mark("compiled");
   std::string body;
   const int64_t action_time = microsecondsOf([&] {
      // the two phases of executeAndOrder back to back (begin validates the ordering and queues the device work)
      const std::unique_ptr<actions::Action::Pending> pending = query.action->begin(database, std::move(filters));
      body = query.action->finishJson(database, *pending);
   });
   Trace::mark("action_done");
   Database::lastTimings() = {filter_time, action_time};
   return body;
}

std::vector<QueryEngine::BatchOutcome> QueryEngine::executeQueries(const std::vector<std::string>& queries, bool render_json) const {
   Trace::reset();
   checkGpu(silo_gpu_set_device(database.device), "silo_gpu_set_device");
   std::vector<BatchOutcome> outcomes(queries.size());
   std::vector<std::unique_ptr<Query>> parsed(queries.size());
   std::vector<std::unique_ptr<actions::Action::Pending>> pending(queries.size());
   actions::ScanBatcher batcher;  // active on this thread until the end of the function
   // phase 1a: parse and compile — and, for the filter -> count queries (Aggregated without groupByFields), lower the filter
   // to its bit-program — queries side by side on the batch workers
   struct CountJob {
      size_t partition;
      OperatorResult filter;                    // shares its state with the copy the action holds
      std::unique_ptr<ProgramBuilder> builder;  // owns the code and leaf arrays the program points into
      silo_gpu_bitprog program;
   };
   std::vector<std::vector<OperatorResult>> filters(queries.size());
   std::vector<std::vector<CountJob>> count_jobs(queries.size());
   const auto parseAndCompile = [&](size_t i) {
      try {
         checkGpu(silo_gpu_set_device(database.device), "silo_gpu_set_device");  // this may be a batch worker's first touch of the device
         parsed[i] = std::make_unique<Query>(queries[i]);
         filters[i] = compileFilter(database, *parsed[i]->filter);
         if (parsed[i]->action->countsOnly()) {
            for (size_t partition_index = 0; partition_index < filters[i].size(); ++partition_index) {
               CountJob job{partition_index, filters[i][partition_index], std::make_unique<ProgramBuilder>(filters[i][partition_index].rows()), {}};
               if (job.filter.prepareCount(*job.builder, job.program)) {
                  if (job.builder->queuedDeviceWork()) {  // the program is launched from the batch's stream, not this thread's
                     checkGpu(silo_gpu_stream_synchronize(queryStream()), "silo_gpu_stream_synchronize");
                  }
                  count_jobs[i].push_back(std::move(job));
               }
            }
         }
      } catch (...) {
         count_jobs[i].clear();
         outcomes[i].error = std::current_exception();
      }
   };
   if (database.broadcast != nullptr) {
      // position-range shards fetch filter leaves from other ranks WHILE compiling: those collectives must be issued in the
      // same order on every rank, so the queries are compiled one after the other
      for (size_t i = 0; i < queries.size(); ++i) {
         parseAndCompile(i);
      }
   } else {
      BatchWorkers::instance().run(queries.size(), parseAndCompile);
   }
   Trace::mark("batch_compiled");
   for (size_t i = 0; i < queries.size(); ++i) {  // phase 1b: validate the actions, queue their scans (this thread's batcher)
      if (outcomes[i].error != nullptr) {
         continue;
      }
      const actions::ScanBatcher::Checkpoint mark = batcher.checkpoint();
      try {
         Database::queryFingerprint() = database.all_reduce != nullptr ? Database::fingerprintOf(queries[i]) : 0;
         pending[i] = parsed[i]->action->begin(database, std::move(filters[i]));
      } catch (...) {
         batcher.rollback(mark);  // scans recorded by the failed query point into buffers that are gone
         count_jobs[i].clear();
         outcomes[i].error = std::current_exception();
      }
   }
   // every filter program of a partition in ONE launch (K3b) instead of one latency-bound launch per query
   std::vector<CountJob*> jobs;
   for (std::vector<CountJob>& of_query : count_jobs) {
      for (CountJob& job : of_query) {
         jobs.push_back(&job);
      }
   }
   Trace::mark("batch_lowered");
   if (jobs.size() > 1) {
      for (size_t partition_index = 0; partition_index < database.partitions.size(); ++partition_index) {
         std::vector<silo_gpu_bitprog> programs;
         std::vector<CountJob*> members;
         for (CountJob* job : jobs) {
            if (job->partition == partition_index) {
               programs.push_back(job->program);
               members.push_back(job);
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
            members[k]->filter.setCount(static_cast<uint32_t>(counts[k]));
         }
      }
   }
   count_jobs.clear();  // a single job takes the ordinary path in finish(): one launch with the count slot
   Trace::mark("batch_queued");
   batcher.flush();  // the scans of all queries, several filters per pass over the planes
   Trace::mark("batch_launched");
   for (size_t i = 0; i < queries.size(); ++i) {  // phase 2: fetch, build rows, order
      if (outcomes[i].error != nullptr) {
         continue;
      }
      try {
         Database::queryFingerprint() = database.all_reduce != nullptr ? Database::fingerprintOf(queries[i]) : 0;
         if (render_json) {  // here rather than after the batch: the device is still busy with the scans of the queries behind this one
            outcomes[i].json = parsed[i]->action->finishJson(database, *pending[i]);
         } else {
            outcomes[i].result = parsed[i]->action->finish(database, *pending[i]);
         }
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
