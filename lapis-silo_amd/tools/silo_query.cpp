// silo_query — minimal stand-alone front end of the engine: load a data set directory in the reference's
// input formats, then answer /query bodies read from stdin (one JSON document per line) with the response
// body the reference's silo_api would send (src/silo_api/query_handler.cpp:22-74), one per line, prefixed by
// the HTTP status.  The HTTP server itself (Poco) is outside the hot path.
//
//   silo_query <dataset directory> [device]  < queries.ndjson
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "silo_engine.h"

int main(int argc, char** argv) {
   if (argc < 2) {
      std::fprintf(stderr, "usage: %s <dataset directory> [device] < queries.ndjson\n", argv[0]);
      return 2;
   }
   const int device = argc > 2 ? std::atoi(argv[2]) : 0;
   silo_engine* engine = nullptr;
   char* summary = nullptr;
   if (silo_engine_create_from_directory(argv[1], device, &engine, &summary) != 0) {
      std::fprintf(stderr, "loading %s failed: %s\n", argv[1], silo_engine_last_error());
      return 1;
   }
   std::fprintf(stderr, "loaded %s: %s\n", argv[1], summary);
   silo_engine_free_string(summary);
   std::string line;
   while (std::getline(std::cin, line)) {
      if (line.find_first_not_of(" \t\r") == std::string::npos) {
         continue;
      }
      char* body = nullptr;
      int status = 0;
      if (silo_engine_execute_query(engine, line.c_str(), &body, &status) != 0) {
         std::fprintf(stderr, "%s\n", silo_engine_last_error());
         return 1;
      }
      int64_t filter_us = 0, action_us = 0;
      silo_engine_last_timings(&filter_us, &action_us);
      std::printf("%d\t%s\n", status, body);
      std::fprintf(stderr, "Execution (filter): %lld microseconds, Execution (action): %lld microseconds\n",
                   static_cast<long long>(filter_us), static_cast<long long>(action_us));
      silo_engine_free_string(body);
   }
   silo_engine_destroy(engine);
   return 0;
}
