// sanitizer_driver.cpp — built with -fsanitize=address,undefined by tests/test_host_sanitizers.py: every golden query
// (and truncated / corrupted variants of it) goes through the JSON parser and the Expression / Action builders, plus the
// insertion and date parsers.  GPU sanitizers are not available on the pool; this covers the host side of the boundary.
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include "database.h"
#include "query_engine.h"
int main(int argc, char** argv) {
   size_t parsed = 0, rejected = 0;
   for (int a = 1; a < argc; ++a) {
      for (const auto& entry : std::filesystem::directory_iterator(argv[a])) {
         std::ifstream in(entry.path());
         std::stringstream buffer; buffer << in.rdbuf();
         const auto document = silo::json::parse(buffer.str());
         const std::string query = document.at("query").dump();
         try { const silo::query_engine::Query q(query); ++parsed; (void)q.filter->toString(silo::Database()); }
         catch (const std::exception&) { ++rejected; }
         // truncated / corrupted variants must fail cleanly
         for (size_t cut = 1; cut < query.size(); cut += 7) {
            try { const silo::query_engine::Query q(query.substr(0, cut)); } catch (const std::exception&) {}
            std::string mutated = query; mutated[cut] = '}';
            try { const silo::query_engine::Query q(mutated); } catch (const std::exception&) {}
         }
      }
   }
   std::cout << "parsed " << parsed << " rejected " << rejected << std::endl;
   silo::storage::column::InsertionColumnPartition column(std::string("main"));
   for (const char* v : {"1:A,2:C", "x", "main:3:G", "::", "1:2:3:4", ""}) { try { column.insert(v, 0); } catch (const std::exception&) {} }
   for (const char* d : {"2021-03-04", "", "x-y-z", "99999999999-1-1", "2021-13-40"}) { (void)silo::common::dateToString(silo::common::stringToDate(d)); }
   return 0;
}
