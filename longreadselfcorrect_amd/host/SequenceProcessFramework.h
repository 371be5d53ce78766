// SequenceProcessFramework.h -- the reference's generic dispatcher (Concurrency/SequenceProcessFramework.h:362-386,
// threaded form :90-230) re-cut as a three-stage pipeline for a device back end.
//
// Same concepts, same guarantees:
//   Processor(const Parameter&);      Output Processor::process(const Input&)             -- classic, per item
//   PostProcessor(const Parameter&);  void PostProcessor::process(const Input&, const Output&)
//       -> called on the calling thread, once per input, in INPUT order, after the batch that contains the item has been
//          processed (reference :183-195); destroyed at the end of the run (its destructor prints the statistics, :385).
//   One Processor instance per worker thread; process() is never called concurrently on one instance (ThreadWorker.h:187-199).
// New: a BATCHED processor concept for device back ends, detected at compile time:
//   std::vector<Output> Processor::process_batch(const std::vector<Input>&)
//   optional  Processor(const Parameter&, size_t worker)            -- which worker (device) this instance serves
//   optional  static size_t Processor::workers(const Parameter&, int thread)   -- how many instances to run (default: thread)
//
// Stages (each on its own threads, bounded queues in between, so that reading, computing and writing overlap the way the
// reference's pthread dispatcher overlaps them):
//   reader      cuts records out of the input block and fills batches (SeqReader is a block parser: this thread is never the limit)
//   workers     one Processor instance each; a worker owns a batch from the queue until its outputs are complete
//   post        the calling thread; consumes finished batches strictly in input order
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "SequenceWorkItem.h"

namespace stride {
namespace SequenceProcessFramework {

const size_t BUFFER_SIZE = 500;        // reference :26

template <class P, class Input, class = void>
struct has_process_batch : std::false_type {};
template <class P, class Input>
struct has_process_batch<P, Input, decltype(void(std::declval<P&>().process_batch(std::declval<const std::vector<Input>&>())))>
    : std::true_type {};

template <class P, class Parameter, class = void>
struct has_workers : std::false_type {};
template <class P, class Parameter>
struct has_workers<P, Parameter, decltype(void(P::workers(std::declval<const Parameter&>(), 0)))> : std::true_type {};

template <class Input, class Output, class Processor>
typename std::enable_if<has_process_batch<Processor, Input>::value, std::vector<Output>>::type
run_batch(Processor& p, const std::vector<Input>& items)
{
    return p.process_batch(items);
}
template <class Input, class Output, class Processor>
typename std::enable_if<!has_process_batch<Processor, Input>::value, std::vector<Output>>::type
run_batch(Processor& p, const std::vector<Input>& items)
{
    std::vector<Output> out;
    out.reserve(items.size());
    for(const Input& it : items) out.push_back(p.process(it));
    return out;
}

template <class Processor, class Parameter>
typename std::enable_if<std::is_constructible<Processor, const Parameter&, size_t>::value, std::unique_ptr<Processor>>::type
make_processor(const Parameter& params, size_t worker) { return std::unique_ptr<Processor>(new Processor(params, worker)); }
template <class Processor, class Parameter>
typename std::enable_if<!std::is_constructible<Processor, const Parameter&, size_t>::value, std::unique_ptr<Processor>>::type
make_processor(const Parameter& params, size_t) { return std::unique_ptr<Processor>(new Processor(params)); }

template <class Processor, class Parameter>
typename std::enable_if<has_workers<Processor, Parameter>::value, size_t>::type
worker_count(const Parameter& params, int thread) { return Processor::workers(params, thread); }
template <class Processor, class Parameter>
typename std::enable_if<!has_workers<Processor, Parameter>::value, size_t>::type
worker_count(const Parameter&, int thread) { return (size_t)(thread > 0 ? thread : 1); }

// processSequences<Input, Output, Processor, PostProcessor, Parameter>(thread, readsFile, params)
template <class Input, class Output, class Processor, class PostProcessor, class Parameter>
size_t processSequences(int thread, const std::string& readsFile, const Parameter& params, size_t batch_items = 0)
{
    const auto t0 = std::chrono::steady_clock::now();
    SeqReader reader(readsFile);
    WorkItemGenerator<Input> generator(&reader);
    PostProcessor postProcessor(params);
    const size_t n_workers = std::max<size_t>(1, worker_count<Processor>(params, thread));
    if(batch_items == 0) batch_items = BUFFER_SIZE;

    struct Batch {
        size_t seq = 0;
        std::vector<Input> items;
        std::vector<Output> outs;
    };
    std::mutex mu;
    std::condition_variable cv_todo, cv_room, cv_done;
    std::deque<std::unique_ptr<Batch>> todo;
    std::map<size_t, std::unique_ptr<Batch>> done;
    // one batch in work per worker + one being filled + one being written: a batch of the default size is ~1 GB of reads plus as
    // much in corrected strings, so the count is what bounds host memory (about 2 GB x (workers + 2))
    const size_t max_in_flight = n_workers + 2;
    size_t in_flight = 0, n_batches = 0;
    bool reading_over = false;

    std::thread reader_thread([&]() {
        size_t seq = 0;
        while(true) {
            std::unique_ptr<Batch> b(new Batch());
            b->seq = seq;
            b->items.reserve(batch_items);
            Input wi;
            while(b->items.size() < batch_items && generator.generate(wi)) b->items.push_back(std::move(wi));
            if(b->items.empty()) break;
            const bool last = b->items.size() < batch_items;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_room.wait(lock, [&]() { return in_flight < max_in_flight; });
                ++in_flight;
                todo.push_back(std::move(b));
            }
            cv_todo.notify_one();
            ++seq;
            if(last) break;
        }
        {
            std::lock_guard<std::mutex> lock(mu);
            reading_over = true;
            n_batches = seq;
        }
        cv_todo.notify_all();
        cv_done.notify_all();
    });

    std::vector<std::thread> workers;
    for(size_t w = 0; w < n_workers; ++w)
        workers.emplace_back([&, w]() {
            std::unique_ptr<Processor> processor = make_processor<Processor>(params, w);
            while(true) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv_todo.wait(lock, [&]() { return !todo.empty() || reading_over; });
                    if(todo.empty()) return;
                    b = std::move(todo.front());
                    todo.pop_front();
                }
                b->outs = run_batch<Input, Output, Processor>(*processor, b->items);
                {
                    std::lock_guard<std::mutex> lock(mu);
                    const size_t seq = b->seq;
                    done[seq] = std::move(b);
                }
                cv_done.notify_all();
            }
        });

    // post-processing on the calling thread, in input order
    size_t next = 0, consumed = 0;
    while(true) {
        std::unique_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> lock(mu);
            cv_done.wait(lock, [&]() { return done.count(next) != 0 || (reading_over && next >= n_batches); });
            auto it = done.find(next);
            if(it == done.end()) break;
            b = std::move(it->second);
            done.erase(it);
        }
        for(size_t i = 0; i < b->items.size(); ++i) postProcessor.process(b->items[i], b->outs[i]);
        consumed += b->items.size();
        {
            std::lock_guard<std::mutex> lock(mu);
            --in_flight;
        }
        cv_room.notify_one();
        ++next;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "Processed %zu sequences (%lfs elapsed)\n", consumed, el);
    }
    reader_thread.join();
    for(std::thread& t : workers) t.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "Processed %zu sequences in %lfs (%lf sequences/s)\n", consumed, secs, (double)consumed / secs);
    return consumed;
}

} // namespace SequenceProcessFramework
} // namespace stride
